// Fused Conv1d on the CDNA4 f16 matrix pipe with fp32-class accuracy: "f16 hi/lo split", 3 products.
//
//   y[b,co,l] = epi( bias[co] + sum_{ci,t} W[co,ci,t] * pro(x)[b,ci, l + t*dil - pad_left] )
//
// Every operand v is carried as two halves  v*s = hi + lo  (hi = f16(v*s), lo = f16(v*s - hi); s a power
// of two that keeps lo in the normal f16 range) and the product is evaluated as
//       hi_w*hi_x + hi_w*lo_x + lo_w*hi_x          (lo_w*lo_x ~ 2^-22 relative: dropped)
// by three v_mfma_f32_32x32x16_f16 into ONE fp32 accumulator.  f16 x f16 products are exact in fp32, so the
// only errors are the 2^-22-relative operand residue and the fp32 accumulation every fp32 conv has anyway:
// measured through the whole decoder the waveform differs from an fp64 evaluation by 2.6e-7 RMS, the fp32
// ATen path by 2.4e-7 (tools/probe_split_precision.py).  Rate: 3 MFMAs at 1024 FLOP/clk/SIMD = 5.3x the
// exact-fp32 MFMA (v_mfma_f32_32x32x2_f32, st2_conv1d.hip) for the same algorithmic FLOPs.
//
// GEMM view per batch item: M = C_out, N = L_out, K = C_in*ks, k ordered (ci/16, tap, ci%16) so that one
// MFMA k-step = 16 consecutive input channels at one tap.
//   * B operand (activations): the workgroup stages the ACTIVATED, split input tile for CI_T channels,
//     [hi|lo][ci/8][BN + (ks-1)*dil positions][8 halves], in LDS -- one ds_read_b128 per fragment, consecutive
//     lanes read consecutive 16-byte slots (conflict free), taps are just a shift of the position index.
//     Double buffered: global loads for chunk c+1 are issued before the MFMAs of chunk c and the prologue
//     (AdaIN affine, Snake / LeakyReLU, split) runs on them afterwards; one barrier per chunk.
//   * A operand (weights): pre-split and packed per load as [ci/16][tap][k-half][co][hi8|lo8]; every wave owns
//     distinct output-channel rows, so its A fragments are read straight from L2 into registers (32 B per
//     lane, prefetched one k-step ahead) and never touch LDS.  The whole packed weight (<= 0.7 MB for the
//     vocoder layers) is L2 resident.
//   * wave tile = 32 (co) x 32*TN (l): TN accumulators of 16 registers.  Waves are arranged WM x WN over
//     (co, l): 4x1 for C_out >= 96, 2x2 for C_out in (32, 96), 1x4 for C_out <= 32.
#pragma once
#include "st2_conv_epilogue.h"
#include <algorithm>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef st2_f32x16 f32x16;

namespace st2f16s {
extern int g_splitk_max, g_splitk_min_chunks;  // st2_conv1d_f16s_set_splitk (st2_conv1d_f16s.hip): 8 slices of >= 4 chunks by default
}

namespace {

constexpr int NT = 256;

struct ChanPar {  // per input channel, staged once per workgroup in LDS (32 B); xs = x_scale, 0 for the channel tail of the last chunk
  float mean, rstd, g, beta, alpha, inv_alpha, xs, pad1;
};

// The one-role kernel's table holds the parameters of channels (2i, 2i + 1) as PAIRS (64 B per pair: the same 32 B per channel), so that
// its prologue runs the affine and the Snake polynomial as packed-f32 ops on two adjacent channels at once (round 6).  On one SIMD
// the prologue's VALU time ADDS to the MFMA time (tools/mfma_valu_overlap.hip) and at C <= 64 it is as large; a v_pk_fma_f32 costs
// ~1.27 x a v_fma_f32 and does the work of two.  Same IEEE operations in the same order per channel: results are bitwise those of
// the scalar form (and of the warp-specialised build, which keeps it).  Plain encodings and low-half broadcasts only
// (tools/check_isa.py: an op_sel on a packed-f32 op would be the gfx950 hazard of DESIGN.md section 9).
typedef float st2_f2 __attribute__((ext_vector_type(2)));
struct ChanPar2 {
  st2_f2 mean, rstd, g, beta, alpha, inv_alpha, xs, pad1;
};

static __device__ __forceinline__ st2_f2 splat2(float c) { return st2_f2{c, c}; }
// sin(x)^2 of st2_act.h's sin_sq, two lanes at a time: the same operations in the same order per component
static __device__ __forceinline__ st2_f2 sin_sq2(st2_f2 x) {
  const st2_f2 n = __builtin_elementwise_rint(x * splat2(0.3183098861837907f));
  st2_f2 r = __builtin_elementwise_fma(n, splat2(-3.1415927410125732f), x);
  r = __builtin_elementwise_fma(n, splat2(8.742277657347586e-08f), r);
  const st2_f2 r2 = r * r;
  st2_f2 p = splat2(2.6000539037340786e-06f);
  p = __builtin_elementwise_fma(p, r2, splat2(-0.00019806614727713168f));
  p = __builtin_elementwise_fma(p, r2, splat2(0.008333017118275166f));
  p = __builtin_elementwise_fma(p, r2, splat2(-0.16666656732559204f));
  const st2_f2 s = __builtin_elementwise_fma(r2 * r, p, r);
  return s * s;
}
static __device__ __forceinline__ st2_f2 snake2(st2_f2 v, st2_f2 alpha, st2_f2 inv_alpha) { return v + inv_alpha * sin_sq2(alpha * v); }

template <int KS, int CI_T, int WM, int WN, int TN>
#ifndef ST2_F16S_OCC
#define ST2_F16S_OCC 2  // workgroups per CU the fused kernel is held to; at 3 (168 VGPRs) every instantiation spills 40-400 B / lane
#endif
__global__ __launch_bounds__(NT, ST2_F16S_OCC) void conv1d_f16s_kernel(const st2_conv_desc d, int* status, int ksplit, float* part) {
  constexpr int BM = 32 * WM;
  constexpr int BN = 32 * TN * WN;
  constexpr int CG = CI_T / 8;    // 8-channel groups per chunk
  constexpr int TPG = NT / CG;    // staging threads per group
  constexpr int S16 = CI_T / 16;  // MFMA k-steps per tap per chunk
  constexpr int MAXXW = BN + (KS - 1) * 8;
  constexpr int R = (MAXXW + TPG - 1) / TPG;  // staging rounds (positions per thread)
  static_assert(WM * WN == 4, "4 waves");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int kg = lane >> 5;
  const int l31 = lane & 31;
  const int wm = wave / WN;
  const int wn = wave % WN;
#ifndef ST2_F16S_DISPATCH_ORDER
  // Workgroups are dealt to the 8 XCDs round-robin by their linear id, so in dispatch order NEIGHBOURING column tiles -- which share
  // the tap halo, a whole 128-B line at each edge of a 1 KB row segment (counters: reads 1.25 x algorithmic at C = 64,
  // profiles/r05/r05w_pmc_narrow.md) -- never share an L2.  Here the column tiles one XCD receives within a row of the grid become a
  // contiguous run: tile index = rank of (xcd, arrival) among the row's tiles.  A bijection on [0, gridDim.x) for any gridDim.x, so
  // results are bitwise those of the dispatch order (-DST2_F16S_DISPATCH_ORDER: tools/build_f16s_xcd.sh builds that library for the
  // A-B); measured -8 % at C = 64 / k = 7, -5 % at k = 11, -8 % at k = 3 / C = 128 with the residual (profiles/r05/r05x_*).
  int bx;
  {
    const int nx = gridDim.x;
    const int o = (int)(((int64_t)nx * (blockIdx.y + (int64_t)gridDim.y * blockIdx.z)) & 7);
    const int t = blockIdx.x + o, xcd = t & 7;
    int before = 0, first_mine = 0;
    for (int r = 0; r < 8; ++r) {
      const int t_first = o + ((r - o) & 7);  // first linear id of this row on XCD r
      const int cnt = t_first < nx + o ? (nx + o - 1 - t_first) / 8 + 1 : 0;
      if (r < xcd) before += cnt;
      if (r == xcd) first_mine = t_first;
    }
    bx = before + (t - first_mine) / 8;
  }
  const int n0 = bx * BN;
#else
  const int n0 = blockIdx.x * BN;
#endif
  const int m0 = blockIdx.y * BM;
  // split-K launches (ksplit > 1): grid.z = B * ksplit, slice `ksl` accumulates the chunks [c_begin, c_end) of the input
  // channels and stores its scaled partial sums to part[ksl][b][co][l]; splitk_reduce_kernel adds the slices in a fixed
  // order and applies the epilogue
  const int b = blockIdx.z / ksplit;
  const int ksl = blockIdx.z - b * ksplit;

  const int XW = BN + (KS - 1) * d.dil;  // staged positions
  // LDS: [2 buffers][2 planes hi/lo][CG][XW] slots of 16 B, then the channel parameter table
  h8* xs = reinterpret_cast<h8*>(smem_raw);
  const int plane = CG * XW;  // slots per plane
  ChanPar2* par2 = reinterpret_cast<ChanPar2*>(smem_raw + (size_t)4 * plane * 16);  // pair i = channels (2i, 2i + 1)

  const int pro = d.pro;
  const bool has_par = pro == ST2_PRO_ADAIN_LEAKY || pro == ST2_PRO_ADAIN_SNAKE || pro == ST2_PRO_SNAKE ||
                       pro == ST2_PRO_COLNORM;
  const int C_pad = (d.C_in + CI_T - 1) / CI_T * CI_T;
  if (has_par) {
    for (int ci = tid; ci < C_pad; ci += NT) {
      ChanPar p = {0.f, 1.f, 1.f, 0.f, 1.f, 1.f, 0.f, 0.f};
      if (ci < d.C_in) {
        p.xs = d.x_scale;
        if (pro == ST2_PRO_COLNORM) {  // per-channel affine of a LayerNorm over channels (statistics are per position)
          const float g = d.gamma[(int64_t)b * d.gb_bs + ci];
          p.g = d.gamma_plus_one ? 1.0f + g : g;
          p.beta = d.beta[(int64_t)b * d.gb_bs + ci];
        } else if (pro != ST2_PRO_SNAKE) {
          const float* st = d.stats + ((int64_t)b * d.C_in + ci) * 2;
          p.mean = st[0];
          p.rstd = st[1];
          p.g = 1.0f + d.gamma[(int64_t)b * d.gb_bs + ci];
          p.beta = d.beta[(int64_t)b * d.gb_bs + ci];
        }
        if (pro == ST2_PRO_ADAIN_SNAKE || pro == ST2_PRO_SNAKE) {
          p.alpha = d.alpha[ci];
          p.inv_alpha = 1.0f / p.alpha;
        }
      }
      float* q = reinterpret_cast<float*>(&par2[ci >> 1]) + (ci & 1);  // component ci & 1 of every pair member
      q[0] = p.mean; q[2] = p.rstd; q[4] = p.g; q[6] = p.beta; q[8] = p.alpha; q[10] = p.inv_alpha; q[12] = p.xs;
    }
  }

  // ---- staging assignment: thread -> (channel group, R positions); identical for every chunk ----------
  const int sg = tid / TPG;
  const int sp0 = tid % TPG;
  const float* xb = d.x + (int64_t)b * d.x_bs;
  const int lin0 = n0 - d.pad_left;  // input position held by staged column 0
  float xr[R][8];

  // loads are unconditional on clamped (always valid) addresses; out-of-range values are zeroed in store_chunk
  auto load_chunk = [&](int c0) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int l = min(max(lin0 + sp0 + r * TPG, 0), d.L_in - 1);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int ci = min(c0 + sg * 8 + e, d.C_in - 1);
        xr[r][e] = xb[(int64_t)ci * d.x_cs + l];
      }
    }
  };

  // prologue + hi/lo split of the R x 8 loaded values, written to LDS buffer `buf`; PRO is a compile-time
  // constant inside so the element loops carry no branches
  auto store_chunk_as = [&](int c0, int buf, auto pro_tag) __attribute__((always_inline)) {
    constexpr int PRO = decltype(pro_tag)::value;
    h8* dst = xs + (size_t)buf * 2 * plane + sg * XW;
    // This thread's 8 channels keep their parameters in REGISTERS across its R positions (round 6): read inside the element loop
    // they were re-read from LDS after every ds_write of a finished slot (the table and the staging buffers share smem_raw: the
    // compiler cannot prove they do not alias) -- 1-3 LDS reads and an lgkmcnt wait per element in a loop whose VALU slots are the
    // kernel's floor (tools/mfma_valu_overlap.hip).  56 VGPRs more (162-189 -> ~206 of 256).  Measured per layout at B = 32
    // (profiles/r06/r06s_probe_narrow.log): the 32-row layout (WM = 1: C <= 32, L = 240 000) gains 5-9 % (k = 7 0.911 -> 0.833 ms, k = 11
    // 0.988 -> 0.941), the 64- and 128-row layouts LOSE 3-8 % -- so only the 32-row layout hoists.
    // (The warp-specialised build holds them in SGPRs -- a wave stages one channel group -- and gains 3-9 %, st2_conv1d_f16s_ws.h;
    // here the same form spills 60-120 SGPRs next to the descriptor and measured 0 ... +5 % SLOWER on every layout, r06z.)
    constexpr bool HOIST = WM == 1;
    // ... and only there the prologue runs packed (two adjacent channels per v_pk_* op, bitwise the scalar form): C = 32 / L = 240 000
    // k = 7 0.924 -> 0.797 ms, k = 11 1.015 -> 0.924 with both; the 64-row layout moves by -5 ... 0 %, the 128-row k = 3 layers of the
    // default step by -6 ... +6 % (profiles/r06/r06w_probe_narrow.log): they keep one channel at a time
    constexpr bool PACK = WM == 1;
    constexpr bool TABLE = PRO == ST2_PRO_ADAIN_LEAKY || PRO == ST2_PRO_ADAIN_SNAKE || PRO == ST2_PRO_SNAKE || PRO == ST2_PRO_COLNORM;
    ChanPar2 pr[4];
    if constexpr (TABLE && HOIST) {
#pragma unroll
      for (int ep = 0; ep < 4; ++ep) pr[ep] = par2[(c0 + sg * 8) / 2 + ep];
    }
    auto par_of = [&](int ep) __attribute__((always_inline)) -> ChanPar2 {
      if constexpr (HOIST) return pr[ep];
      else return par2[(c0 + sg * 8) / 2 + ep];
    };
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int pos = sp0 + r * TPG;
      if (pos >= XW) continue;
      const int l = lin0 + pos;
      const bool lok = l >= 0 && l < d.L_in;
      float cmean = 0.f, crstd = 1.f;
      if constexpr (PRO == ST2_PRO_COLNORM) {
        const float* st = d.stats + ((int64_t)b * d.L_in + min(max(l, 0), d.L_in - 1)) * 2;
        cmean = st[0];
        crstd = st[1];
      }
      h8 hi, lo;
      bool sat = false;
#pragma unroll
      for (int ep = 0; ep < 4; ++ep) {  // channels (2 ep, 2 ep + 1) of this thread's group of 8, as one packed pair
        const int ci = c0 + sg * 8 + 2 * ep;
        st2_f2 v = st2_f2{xr[r][2 * ep], xr[r][2 * ep + 1]};
        if constexpr (PACK) {
          if constexpr (PRO == ST2_PRO_LEAKY) {
            v = st2_f2{leaky(v.x, d.slope), leaky(v.y, d.slope)};
          } else if constexpr (PRO == ST2_PRO_ADAIN_LEAKY) {
            const ChanPar2 p = par_of(ep);
            st2_f2 u = (v - p.mean) * p.rstd;
            u = p.g * u + p.beta;
            v = st2_f2{leaky(u.x, d.slope), leaky(u.y, d.slope)};
          } else if constexpr (PRO == ST2_PRO_ADAIN_SNAKE) {
            const ChanPar2 p = par_of(ep);
            st2_f2 u = (v - p.mean) * p.rstd;
            u = p.g * u + p.beta;
            v = snake2(u, p.alpha, p.inv_alpha);
          } else if constexpr (PRO == ST2_PRO_SNAKE) {
            const ChanPar2 p = par_of(ep);
            v = snake2(v, p.alpha, p.inv_alpha);
          } else if constexpr (PRO == ST2_PRO_COLNORM) {
            const ChanPar2 p = par_of(ep);
            // per-position statistics stay scalar: splatting crstd -- the HIGH element of the (mean, rstd) pair its load returns --
            // would be an op_sel'd packed op (tools/check_isa.py)
            const st2_f2 u = st2_f2{(v.x - cmean) * crstd, (v.y - cmean) * crstd};
            v = u * p.g + p.beta;
          }
          if constexpr (TABLE)
            v = v * par_of(ep).xs;
          else
            v = v * splat2(d.x_scale);
        } else {  // one channel at a time (the 64- / 128-row layouts: the packed form is neutral to slower there, r06w)
          [[maybe_unused]] ChanPar2 p;
          if constexpr (TABLE) p = par_of(ep);
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            float w = v[c];
            if constexpr (PRO == ST2_PRO_LEAKY) {
              w = leaky(w, d.slope);
            } else if constexpr (PRO == ST2_PRO_ADAIN_LEAKY) {
              float u = (w - p.mean[c]) * p.rstd[c];
              u = p.g[c] * u + p.beta[c];
              w = leaky(u, d.slope);
            } else if constexpr (PRO == ST2_PRO_ADAIN_SNAKE) {
              float u = (w - p.mean[c]) * p.rstd[c];
              u = p.g[c] * u + p.beta[c];
              w = snake(u, p.alpha[c], p.inv_alpha[c]);
            } else if constexpr (PRO == ST2_PRO_SNAKE) {
              w = snake(w, p.alpha[c], p.inv_alpha[c]);
            } else if constexpr (PRO == ST2_PRO_COLNORM) {
              const float u = (w - cmean) * crstd;
              w = u * p.g[c] + p.beta[c];
            }
            if constexpr (TABLE)
              w = w * p.xs[c];
            else
              w = w * d.x_scale;
            v[c] = w;
          }
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int e = 2 * ep + c;
          float w = c ? v.y : v.x;
          if constexpr (TABLE)
            w = lok ? w : 0.f;
          else
            w = (lok && ci + c < d.C_in) ? w : 0.f;
          const float vc = st2_clamp_f16(w);  // saturate instead of inf / NaN, reported via st2_status()
          sat |= vc != w;
          const _Float16 h = (_Float16)vc;
          hi[e] = h;
          lo[e] = (_Float16)(vc - (float)h);
        }
      }
      if (sat) st2_raise_status(status, ST2_STATUS_F16_RANGE);
      dst[pos] = hi;
      dst[plane + pos] = lo;
    }
  };
  auto store_chunk = [&](int c0, int buf) __attribute__((always_inline)) {
    switch (pro) {
      case ST2_PRO_LEAKY:
        store_chunk_as(c0, buf, std::integral_constant<int, ST2_PRO_LEAKY>{});
        break;
      case ST2_PRO_ADAIN_LEAKY:
        store_chunk_as(c0, buf, std::integral_constant<int, ST2_PRO_ADAIN_LEAKY>{});
        break;
      case ST2_PRO_ADAIN_SNAKE:
        store_chunk_as(c0, buf, std::integral_constant<int, ST2_PRO_ADAIN_SNAKE>{});
        break;
      case ST2_PRO_SNAKE:
        store_chunk_as(c0, buf, std::integral_constant<int, ST2_PRO_SNAKE>{});
        break;
      case ST2_PRO_COLNORM:
        store_chunk_as(c0, buf, std::integral_constant<int, ST2_PRO_COLNORM>{});
        break;
      default:
        store_chunk_as(c0, buf, std::integral_constant<int, ST2_PRO_NONE>{});
        break;
    }
  };

  f32x16 acc[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  // ---- A operand stream: 32 B (hi8|lo8) per lane per k-step, constant stride between steps -----------
  const int co_a = m0 + wm * 32 + l31;  // < wq_co_pad by construction of the packing
  const h8* ap = reinterpret_cast<const h8*>(d.wq) + ((int64_t)kg * d.wq_co_pad + co_a) * 2;
  const int64_t a_step = (int64_t)2 * d.wq_co_pad * 2;  // h8 units per k-step
  const int nchunk_all = C_pad / CI_T;
  const int c_begin = (int)((int64_t)ksl * nchunk_all / ksplit);
  const int nchunk = (int)((int64_t)(ksl + 1) * nchunk_all / ksplit) - c_begin;  // chunks of this slice (>= 1)
  constexpr int SPC = S16 * KS;  // k-steps per chunk
  ap += (int64_t)c_begin * SPC * a_step;

  load_chunk(c_begin * CI_T);
  if (has_par) __syncthreads();  // parameter table visible
  store_chunk(c_begin * CI_T, 0);
  // weight fragments are double buffered in two NAMED register sets indexed by the (compile-time) parity of the
  // k-step inside the chunk; with one set hipcc re-uses the registers and sinks the prefetch to ~4 MFMAs ahead
  // of its consumer
  h8 a_hi[2], a_lo[2];
  a_hi[0] = ap[0];
  a_lo[0] = ap[1];
  __syncthreads();

  // Workgroups sharing a CU run out of phase: while this wave is in its k loop, a neighbour's may be in its epilogue
  // (VALU + global memory).  Raised priority for the k loop keeps the matrix pipe fed first (cdna_hip_programming.md
  // T5: pays where waves have different roles); dropped again before the epilogue.
  __builtin_amdgcn_s_setprio(1);
  for (int c = 0; c < nchunk; ++c) {
    const int buf = c & 1;
    const bool more = c + 1 < nchunk;
    const h8* xbuf = xs + (size_t)buf * 2 * plane + kg * XW + wn * (32 * TN) + l31;
#pragma unroll
    for (int s = 0; s < S16; ++s) {
#pragma unroll
      for (int t = 0; t < KS; ++t) {
        const int cur = (s * KS + t) & 1, nxt = cur ^ 1;
        // All VMEM below is issued unconditionally (the very last prefetch re-reads the last fragment, the last
        // chunk's activation loads clamp to the last channel): a branch around a load makes hipcc's in-order vmcnt
        // bookkeeping conservative, and the next weight wait then drains the activation loads at HBM latency.
        if (more || s + 1 < S16 || t + 1 < KS) ap += a_step;
        a_hi[nxt] = ap[0];  // prefetch the next k-step's weights
        a_lo[nxt] = ap[1];
        // next chunk's activations: issued AFTER the weight prefetch so that the in-order vmcnt wait of the next
        // k-step does not have to drain these (possibly HBM-latency) loads
        if (s == 0 && t == 0) load_chunk((c_begin + c + 1) * CI_T);
        __builtin_amdgcn_sched_barrier(0x786);  // neither VMEM nor MFMA crosses: the prefetch stays a full k-step ahead
        const h8 ah = a_hi[cur], al = a_lo[cur];
        const h8* xp = xbuf + (2 * s) * XW + t * d.dil;
        h8 bh[TN], bl[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          bh[j] = xp[j * 32];
          bl[j] = xp[plane + j * 32];
        }
#pragma unroll
        // activations as the MFMA's A operand, weights as B: the accumulator is the transposed tile the shared
        // epilogue expects (st2_conv_epilogue.h): lane = output row, registers = runs of 4 consecutive positions
        for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], ah, acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[j], ah, acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], al, acc[j], 0, 0, 0);
      }
    }
    if (SPC & 1) {  // odd step count: next chunk's step 0 reads set 0
      a_hi[0] = a_hi[1];
      a_lo[0] = a_lo[1];
    }
    if (more) store_chunk((c_begin + c + 1) * CI_T, buf ^ 1);
    __syncthreads();
  }

  __builtin_amdgcn_s_setprio(0);
  if (ksplit > 1) {
    // Partial sums of this K slice, scaled, in the ACCUMULATOR'S OWN layout: [slice][b][tile y][tile x][q = 4 j + r / 4][thread] x
    // float4 -- every store instruction of a wave is one contiguous KB.  (Until round 6 the slices were dense [C_out][L_out]
    // images: with a lane per output row that is 64 four-byte stores to 64 different lines per instruction, 64 instructions per
    // wave -- 7 of the 17 us such a launch took at one utterance.)
    const float* rsc1 = d.w_row_scale ? d.w_row_scale : reinterpret_cast<const float*>(d.wq);
    const int row = m0 + wm * 32 + l31;  // transposed accumulator: this lane's output row
    const float sraw = rsc1[row];
    const float osc_r = d.w_row_scale ? d.out_scale * sraw : d.out_scale;
    float4* pb = reinterpret_cast<float4*>(part) +
                 ((((int64_t)ksl * d.B + b) * gridDim.y + blockIdx.y) * gridDim.x + n0 / BN) * (TN * 4) * NT + tid;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        pb[(j * 4 + q) * NT] = float4{acc[j][4 * q] * osc_r, acc[j][4 * q + 1] * osc_r, acc[j][4 * q + 2] * osc_r,
                                      acc[j][4 * q + 3] * osc_r};
    }
    return;
  }
  // ---- epilogue (st2_conv_epilogue.h: shared with the xs kernel; emits the InstanceNorm partial sums on request) ----
  st2_conv_epilogue<TN, WM, WN>(d, acc, b, m0, n0, wm, wn, l31, kg);
}

// Second half of a split-K conv: y = epi(bias + sum over the K slices, in slice order).  One thread per float4 of a slice (four
// consecutive positions of one output row), addressed exactly as the conv kernel's thread `threadIdx.x` of tile (blockIdx.y) stored
// its accumulator group q = blockIdx.x: WN / TN are the tile's wave grid and column tiles per wave.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const st2_conv_desc d, int ksplit, const float* part, int WN, int TN,
                                                            int tiles_x) {
  const int tid = threadIdx.x, q = blockIdx.x, tile = blockIdx.y, b = blockIdx.z;
  const int wave = tid >> 6, lane = tid & 63, kg = lane >> 5, l31 = lane & 31;
  const int wm = wave / WN, wn = wave % WN;
  const int by = tile / tiles_x, bx = tile - by * tiles_x;
  const int co = by * (128 / WN) + wm * 32 + l31;  // BM = 32 * WM, WM * WN = 4
  const int l0 = bx * (32 * TN * WN) + wn * (32 * TN) + (q >> 2) * 32 + 8 * (q & 3) + 4 * kg;
  if (co >= d.C_out || l0 >= d.L_out) return;
  const int64_t slice = (int64_t)d.B * gridDim.y * (TN * 4) * 256;  // float4 per slice
  const float4* p = reinterpret_cast<const float4*>(part) + (((int64_t)b * gridDim.y + tile) * (TN * 4) + q) * 256 + tid;
  float4 a4 = p[0];
  for (int s = 1; s < ksplit; ++s) {
    const float4 t = p[s * slice];
    a4.x += t.x; a4.y += t.y; a4.z += t.z; a4.w += t.w;
  }
  const float av[4] = {a4.x, a4.y, a4.z, a4.w};
  const float bias = d.bias ? d.bias[co] : 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int l = l0 + c;
    if (l >= d.L_out) break;
    float v = av[c] + bias;
    if (d.res) v += d.res[(int64_t)b * d.res_bs + (int64_t)co * d.res_cs + (l >> d.res_shift)];
    if (d.res2) v = d.res2[(int64_t)b * d.res2_bs + (int64_t)co * d.res2_cs + l] + v;
    if (d.div != 1.0f) v = v / d.div;
    switch (d.act) {
      case ST2_ACT_GELU: v = gelu_erf(v); break;
      case ST2_ACT_EXP_SIN: v = co < d.act_split ? expf(v) : sin_acc(v); break;
      case ST2_ACT_TANH: v = tanhf(v); break;
      case ST2_ACT_LEAKY: v = leaky(v, d.act_slope); break;
      case ST2_ACT_GELU_TANH: v = gelu_tanh(v); break;
      default: break;
    }
    d.y[(int64_t)b * d.y_bs + (int64_t)co * d.y_cs + l] = v;
  }
}

// K slices of a launch: layers whose grid leaves most of the chip idle AND whose k loop is long run as `ksplit`
// workgroups per tile + a reduction (a 1024 -> 2048 Linear over 100 tokens is 8-16 workgroups walking 32-64 chunks in
// series, ~70 us; at B = 1 these launches are half of a sentence's time).  A function of the geometry only, so every
// plan picks the same split (bitwise-equal results); 1 when the caller provides no (or too small a) workspace.
inline int ksplit_for_geometry(const st2_conv_desc& d) {
  const int CI_T = d.ks <= 3 ? 32 : 16;
  const int BM = d.C_out > 64 ? 128 : (d.C_out > 32 ? 64 : 32);
  const int BN = d.C_out > 64 ? 128 : (d.C_out > 32 ? 256 : 512);
  const int nchunk = (d.C_in + CI_T - 1) / CI_T;
  const int64_t wgs = (int64_t)st2_cdiv(d.L_out, BN) * st2_cdiv(d.C_out, BM) * d.B;
  // measured (profiles/archive/r02/r02v_*, r02w_*): with 256 workgroups (B = 32 x 8 co-blocks) a 4-way split LOSES 4-6 % on the
  // LibriTTS configurations (the reduction re-reads 4 x the output), with 8-16 (B = 1) an 8-way split takes the
  // long-form passage from 181 to 118 ms -- so only launches below half a round of the chip are split, up to ~256 slices
  if (nchunk < 8 || wgs >= 128) return 1;
  int s = (int)std::min<int64_t>(st2f16s::g_splitk_max, 256 / wgs);
  s = std::min(s, nchunk / st2f16s::g_splitk_min_chunks);
  return std::max(s, 1);
}
// bytes of `s` slices: whole tiles (the slices are stored in the accumulator layout, splitk_reduce_kernel)
inline int64_t splitk_bytes_for(const st2_conv_desc& d, int s) {
  const int BM = d.C_out > 64 ? 128 : (d.C_out > 32 ? 64 : 32);
  const int BN = d.C_out > 64 ? 128 : (d.C_out > 32 ? 256 : 512);
  return (int64_t)s * d.B * st2_cdiv(d.C_out, BM) * BM * st2_cdiv(d.L_out, BN) * BN * 4;
}
inline int pick_ksplit(const st2_conv_desc& d) {
  const int s = ksplit_for_geometry(d);
  if (s <= 1 || !d.splitk_ws || splitk_bytes_for(d, s) > d.splitk_ws_bytes) return 1;
  return s;
}

template <int KS, int CI_T, int WM, int WN, int TN>
int launch(const st2_conv_desc& d, hipStream_t s) {
  constexpr int BM = 32 * WM;
  constexpr int BN = 32 * TN * WN;
  const int XW = BN + (KS - 1) * d.dil;
  const int C_pad = (d.C_in + CI_T - 1) / CI_T * CI_T;
  const bool has_par = d.pro == ST2_PRO_ADAIN_LEAKY || d.pro == ST2_PRO_ADAIN_SNAKE || d.pro == ST2_PRO_SNAKE ||
                       d.pro == ST2_PRO_COLNORM;
  const size_t smem = (size_t)4 * (CI_T / 8) * XW * 16 + (has_par ? (size_t)C_pad * 32 : 0);
  ST2_REQUIRE(smem <= 160 * 1024, "st2_conv1d_f16s: tile needs %zu B of LDS (ks=%d dil=%d C_in=%d)", smem, KS,
              d.dil, d.C_in);
  ST2_REQUIRE(d.wq_cin_pad == C_pad, "st2_conv1d_f16s: packed weight has %d input channels, kernel needs %d",
              d.wq_cin_pad, C_pad);
  ST2_REQUIRE(d.wq_co_pad % BM == 0 && d.wq_co_pad >= d.C_out, "st2_conv1d_f16s: wq_co_pad=%d must be a multiple "
              "of %d covering C_out=%d", d.wq_co_pad, BM, d.C_out);
  static std::atomic<uint64_t> attr_done{0};  // one bit per device ordinal
  st2_once_per_device(attr_done, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1d_f16s_kernel<KS, CI_T, WM, WN, TN>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  if (d.part) ST2_REQUIRE(d.part_nt >= st2_cdiv(d.L_out, 128), "st2_conv1d_f16s: part_nt=%d < %d tiles", d.part_nt,
                          st2_cdiv(d.L_out, 128));
  const int ksplit = d.part ? 1 : pick_ksplit(d);  // the reduction kernel does not emit statistics
  ST2_REQUIRE((int64_t)d.B * ksplit <= 65535, "st2_conv1d_f16s: grid too large");
  // rows beyond C_out inside the last co block are computed on zero weights and not stored
  dim3 grid(st2_cdiv(d.L_out, BN), st2_cdiv(d.C_out, BM), d.B * ksplit);
  float* part = ksplit > 1 ? reinterpret_cast<float*>(d.splitk_ws) : nullptr;
  hipLaunchKernelGGL((conv1d_f16s_kernel<KS, CI_T, WM, WN, TN>), grid, dim3(NT), smem, s, d, st2_status_device_ptr(),
                     ksplit, part);
  ST2_CHECK_LAUNCH("st2_conv1d_f16s");
  if (ksplit > 1) {
    static_assert(NT == 256, "splitk_reduce_kernel mirrors a 256-thread tile");
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(TN * 4, grid.x * grid.y, d.B), dim3(256), 0, s, d, ksplit, part, WN, TN,
                       (int)grid.x);
    ST2_CHECK_LAUNCH("st2_conv1d_f16s (split-K reduction)");
  }
  return 0;
}

}  // namespace

namespace st2ws {  // st2_conv1d_f16s_ws.h, instantiated in st2_conv1d_f16s_w{0,1,2}.hip
template <int KS, int CI_T>
int launch_ws_by_cout(const st2_conv_desc& d, hipStream_t s);
}

namespace st2f16s {

extern int g_variant;  // st2_conv1d_f16s_set_variant: 0 = rule, 1 = one role per wave always, 2 = warp-specialised when eligible

// The warp-specialised persistent build (st2_conv1d_f16s_ws.h; bitwise the same results) exists for the vocoder's AdaIN + Snake
// convs with k = 3 / 7 / 11 and C_out <= 64.  BY RULE it takes the k = 3 ones whose grid gives every CU at least two tiles:
// x1.12-1.20 there (tools/probe_ws.py; inside the HiFi-GAN configuration 0.82 against 1.01 ms at C = 32, L = 240 000).  At
// k = 7 / 11 it is within +-7 % of the one-role kernel and loses 17 % at dilation 1, at C_out = 128 it loses 9-18 %
// (profiles/archive/r03/r03A_probe_ws.log): those stay on the one-role kernel; st2_conv1d_f16s_set_variant(2) forces every eligible layer.
template <int KS>
inline bool ws_eligible(const st2_conv_desc& d) {
  if constexpr (KS != 3 && KS != 7 && KS != 11) return false;
  if (d.pro != ST2_PRO_ADAIN_SNAKE || g_variant == 1 || d.C_in > 256 || d.C_out > 64) return false;
  if (ksplit_for_geometry(d) > 1) return false;
  if (g_variant == 2) return true;
  const int BM = d.C_out > 32 ? 64 : 32;
  const int BN = d.C_out > 32 ? 256 : 512;
  return KS == 3 && (int64_t)st2_cdiv(d.L_out, BN) * st2_cdiv(d.C_out, BM) * d.B >= 512;
}

template <int KS, int CI_T>
int launch_by_cout(const st2_conv_desc& d, hipStream_t s) {
  if constexpr (KS == 3 || KS == 7 || KS == 11) {
    if (ws_eligible<KS>(d)) return st2ws::launch_ws_by_cout<KS, CI_T>(d, s);
  }
  if (d.C_out > 64) return launch<KS, CI_T, 4, 1, 4>(d, s);  // 128 co x 128 l
  if (d.C_out > 32) return launch<KS, CI_T, 2, 2, 4>(d, s);  // 64 co x 256 l
  return launch<KS, CI_T, 1, 4, 4>(d, s);                    // 32 co x 512 l
}

}  // namespace st2f16s
