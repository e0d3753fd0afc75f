// Toom-Cook F(M, R) form of the split-f16 MFMA Conv1d: device kernels (transform pass + conv over tiles) and their argument
// structs.  EXPERIMENTAL -- included by tools/wino_bench.hip only (stand-alone harness with an fp64 checker and a GPU-free
// selftest); not compiled into libst2_hip.so and not measured on a GPU yet.  Kept under csrc/ so
// that a library translation unit can include the same source once the experiment has been timed.
//
// Maths (P = M + R - 1 points; matrices built on the host, see tools/wino_bench.hip toom()):
//   y[M T + i] = sum_j w[j] a[M T + i + j - pad],  w zero-padded to R G taps, group g = taps R g .. R g + R - 1
//   V_p[ci][T'] = sum_n BT[p][n] a[ci][M T' - pad + n]               (input transform, fp32, then hi / lo f16 split)
//   U_{g,p}[co][ci] = sum_r Gm[p][r] w[co][ci][R g + r]              (weight transform, fp64 at pack time)
//   Y_p[co][T]  = sum_{g, ci} U_{g,p}[co][ci] V_p[ci][T + g]         (P independent G-tap convs over the TILE index: MFMA)
//   y[co][M T + i] = sum_p AT[i][p] Y_p[co][T]                       (inverse transform on the fp32 accumulators)
// Kernel structure = st2_conv1d_xs_impl.h with "taps" t = g * P + p: the packed-weight layout (st2.h) is reused with
// ks_eff = P G, the chunk image in LDS has one row set per point, the accumulators are acc[p][j] (transposed tile: lane =
// output row, registers = runs of 4 consecutive TILES = 4 M consecutive outputs -> M 16-byte stores).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "../styletts2_amd/csrc/st2_act.h"

namespace st2w {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int M_, int R_>
struct Scheme {
  static constexpr int M = M_, R = R_, P = M_ + R_ - 1;  // outputs per tile (= tile step), taps per group, points
};
typedef Scheme<3, 3> S33;
typedef Scheme<4, 4> S44;
typedef Scheme<6, 6> S66;
struct Toom {  // transform matrices in fp32, sized for the largest scheme (kernel arguments: scalar loads)
  float BT[11][11];  // V_p = sum_n BT[p][n] d[n]
  float AT[6][11];   // y_i = sum_p AT[i][p] Y_p
};
constexpr int NT = 256;     // threads per workgroup
constexpr int CI_T = 16;    // input channels per chunk (one MFMA k-step per (g, p))
constexpr int CG = CI_T / 8;

struct WArgs {
  // transformed activation planes: vs[b][plane (hi, lo)][p][cg][Lt] slots of 8 channels x f16
  const h8* vs; int cg_tot; int Lt;
  // packed weights, st2.h layout with ks_eff = 5 G: [(i16 * ks_eff + t) * 2 + kg][co_pad][hi8 | lo8]
  const h8* wq; int co_pad, cin_pad;
  const float* row_scale; const float* bias;
  float out_scale;
  float* y; int64_t y_bs; int y_cs;
  const float* res; int64_t res_bs; int res_cs;
  const float* res2; int64_t res2_bs; int res2_cs;  // second residual (the MRF accumulator), may be null
  float div;                                        // final division (number of MRF branches); 0 or 1 = none
  float* part; int part_nt;   // per (b, co, tile block of 96 TN outputs): (sum, sum of squares) of the stored values
  int C_out, L_out;
  Toom tm;
  // dilation d > 1: the grid's batch index is the VIRTUAL batch vb = b * d + r, one per residue r of l = d q + r; the kernel
  // convolves the stride-d subsequence a_r[q] = a[d q + r] (planes and output are per virtual batch: the output tensor is
  // residue-major [b][r][co][q], which the next activation pass un-permutes) and L_out is the natural length
  int dil;
};

template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// conv over tiles: workgroup = 4 waves as WM (co blocks of 32) x WN = 4 / WM (blocks of 32 TN tiles); waves that share their
// output rows (WN > 1) read the same weight fragments in step, i.e. mostly from the CU's vector L1 instead of L2
// ---------------------------------------------------------------------------------------------------------------------
// OCC = workgroups per CU the register budget is held to (3: <= 168 VGPRs; F(4,4) then parks the staged activations of the
// next chunk in scratch -- 3 x 16 B per lane per chunk of ~2000 MFMA cycles)
template <class S, int G, int TN, int OCC, int WM>
__global__ __launch_bounds__(NT, OCC) void conv_w3_kernel(const WArgs d) {
  constexpr int P = S::P, M = S::M, WN = 4 / WM, BM = 32 * WM;
  constexpr int BT_ = 32 * TN * WN;       // tiles per workgroup
  constexpr int XW = BT_ + G - 1;         // staged tile slots per image row
  constexpr int ROWS = 2 * P * CG;        // image rows per chunk: (plane, point, channel group)
  constexpr int SLOTS = ROWS * XW;
  constexpr int NS = (SLOTS + NT - 1) / NT;
  constexpr int LBUF = NS * NT;
  constexpr int SPC = G * P;              // k-steps per chunk
  constexpr int NSET = 3;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  h8* lds = reinterpret_cast<h8*>(smem_raw);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int kg = lane >> 5;
  const int l31 = lane & 31;
  const int t0 = blockIdx.x * BT_;        // first tile
  const int m0 = blockIdx.y * BM;
  const int b = blockIdx.z;
  const int L_eff = d.dil > 1 ? (d.L_out - (b % d.dil) + d.dil - 1) / d.dil : d.L_out;  // outputs of this (virtual) batch item

  const int64_t pstride = (int64_t)d.cg_tot * d.Lt;      // slots per (plane, point)
  const int64_t plane_stride = (int64_t)P * pstride;     // slots per plane
  const h8* vsb = d.vs + (int64_t)b * 2 * plane_stride + t0;
  int soff[NS];
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const int slot = tid + i * NT;
    const int row = slot / XW;
    const int col = slot - row * XW;
    const int pl = row / (P * CG), rem = row % (P * CG), p = rem / CG, g8 = rem % CG;
    soff[i] = slot < SLOTS ? (int)(pl * plane_stride + p * pstride + (int64_t)g8 * d.Lt + col) : 0;
  }
  h8 xr[NS];
  auto load_chunk = [&](int c) __attribute__((always_inline)) {
    const h8* src = vsb + (int64_t)c * CG * d.Lt;
#pragma unroll
    for (int i = 0; i < NS; ++i) xr[i] = src[soff[i]];
  };
  auto store_chunk = [&](int buf) __attribute__((always_inline)) {
    h8* dst = lds + (size_t)buf * LBUF;
#pragma unroll
    for (int i = 0; i < NS; ++i) dst[tid + i * NT] = xr[i];
  };

  f32x16 acc[P][TN];
#pragma unroll
  for (int p = 0; p < P; ++p)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[p][j][r] = 0.f;

  const int co_a = m0 + wm * 32 + l31;  // < co_pad by construction of the packing
  const h8* ap = d.wq + ((int64_t)kg * d.co_pad + co_a) * 2;
  const int64_t a_step = (int64_t)2 * d.co_pad * 2;
  const int nchunk = d.cin_pad / CI_T;
  const int nsteps = nchunk * SPC;

  load_chunk(0);
  h8 a_hi[NSET], a_lo[NSET];
#pragma unroll
  for (int k = 0; k < NSET - 1; ++k) {
    if (k > 0 && k < nsteps) ap += a_step;
    a_hi[k] = ap[0];
    a_lo[k] = ap[1];
  }
  store_chunk(0);
  __syncthreads();

  __builtin_amdgcn_s_setprio(1);
  for (int c = 0; c < nchunk; ++c) {
    const int buf = c & 1;
    const bool more = c + 1 < nchunk;
    const h8* xbuf = lds + (size_t)buf * LBUF + wn * (32 * TN) + l31;
    static_for<SPC>([&](auto i_tag) __attribute__((always_inline)) {
      constexpr int i = decltype(i_tag)::value;
      constexpr int g = i / P, p = i % P;
      constexpr int cur = i % NSET, pre = (i + NSET - 1) % NSET;
      if (more || i + NSET - 1 < SPC) ap += a_step;
      a_hi[pre] = ap[0];
      a_lo[pre] = ap[1];
      if constexpr (i == 0) load_chunk(more ? c + 1 : c);
      if constexpr (i == SPC - 1) store_chunk(buf ^ 1);
      __builtin_amdgcn_sched_barrier(0x786);
      const h8 ah = a_hi[cur], al = a_lo[cur];
      // image rows of point p: hi plane row (p * CG + kg), lo plane row ((P + p) * CG + kg); a tap group is a shift by g tiles
      const h8* xp = xbuf + (p * CG + kg) * XW + g;
      h8 bh[TN], bl[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bh[j] = xp[j * 32];
        bl[j] = xp[P * CG * XW + j * 32];
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[p][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], ah, acc[p][j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[p][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[j], ah, acc[p][j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[p][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], al, acc[p][j], 0, 0, 0);
    });
    if constexpr (SPC % NSET != 0) {  // the next chunk indexes its steps from 0 again: rotate the live sets
      h8 th[NSET], tl[NSET];
#pragma unroll
      for (int k = 0; k < NSET; ++k) {
        th[k] = a_hi[k];
        tl[k] = a_lo[k];
      }
#pragma unroll
      for (int k = 0; k < NSET; ++k) {
        a_hi[k] = th[(k + SPC) % NSET];
        a_lo[k] = tl[(k + SPC) % NSET];
      }
    }
    __syncthreads();
  }
  __builtin_amdgcn_s_setprio(0);

  // ---- epilogue: inverse transform, scale, bias, residual, store, statistics ------------------------------------------
  // lane (l31, kg) owns output row co and, in acc[p][j][4 q + e], tile T = tw + 32 j + 8 q + 4 kg + e: the four tiles of a
  // (j, q) are 4 M consecutive outputs starting at M (tw + 32 j + 8 q + 4 kg)
  const int tw = t0 + wn * (32 * TN);  // this wave's first tile
  const int co = m0 + wm * 32 + l31;
  const bool rok = co < d.C_out;
  const int coc = rok ? co : d.C_out - 1;
  const float osc_r = d.out_scale * d.row_scale[co];
  const float bias_r = d.bias ? d.bias[coc] : 0.f;
  float* yb = d.y + (int64_t)b * d.y_bs + (int64_t)coc * d.y_cs;
  const float* rb = d.res ? d.res + (int64_t)b * d.res_bs + (int64_t)coc * d.res_cs : nullptr;
  const float* r2b = d.res2 ? d.res2 + (int64_t)b * d.res2_bs + (int64_t)coc * d.res2_cs : nullptr;
  const bool use_div = d.div != 0.f && d.div != 1.0f;
  const bool vec_ok = ((reinterpret_cast<uintptr_t>(d.y) | (uintptr_t)(d.y_bs * 4) | (uintptr_t)(d.y_cs * 4)) & 15) == 0 &&
                      (!d.res || ((reinterpret_cast<uintptr_t>(d.res) | (uintptr_t)(d.res_bs * 4) | (uintptr_t)(d.res_cs * 4)) & 15) == 0) &&
                      (!d.res2 || ((reinterpret_cast<uintptr_t>(d.res2) | (uintptr_t)(d.res2_bs * 4) | (uintptr_t)(d.res2_cs * 4)) & 15) == 0);
  const bool full = vec_ok && m0 + BM <= d.C_out && M * (t0 + BT_) <= L_eff;  // workgroup-uniform
  float s1 = 0.f, s2 = 0.f;
  static_for<TN * 4>([&](auto jq_tag) __attribute__((always_inline)) {
    constexpr int j = decltype(jq_tag)::value / 4, q = decltype(jq_tag)::value % 4;
    const int l0 = M * (tw + 32 * j + 8 * q + 4 * kg);
    float o[4 * M];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int i = 0; i < M; ++i) {
        float t = d.tm.AT[i][0] * acc[0][j][4 * q + e];
#pragma unroll
        for (int p = 1; p < P; ++p) t = fmaf(d.tm.AT[i][p], acc[p][j][4 * q + e], t);
        o[M * e + i] = t;
      }
    if (full) {
      f32x4 rv[M];
      if (rb) {
#pragma unroll
        for (int v = 0; v < M; ++v) rv[v] = *reinterpret_cast<const f32x4*>(rb + l0 + 4 * v);
      }
#pragma unroll
      for (int v = 0; v < M; ++v) {
        f32x4 w, r2v;  // the MRF accumulator (one conv in eighteen) is fetched per store instead of held for the whole quad
        if (r2b) r2v = *reinterpret_cast<const f32x4*>(r2b + l0 + 4 * v);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = fmaf(o[4 * v + e], osc_r, bias_r);  // the order of st2_conv_epilogue.h: bias, residual, MRF sum, divide
          if (rb) t += rv[v][e];
          if (r2b) t = r2v[e] + t;
          if (use_div) t = t / d.div;
          w[e] = t;
          s1 += t;
          s2 = fmaf(t, t, s2);
        }
        *reinterpret_cast<f32x4*>(yb + l0 + 4 * v) = w;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4 * M; ++e) {
        const int l = l0 + e;
        const bool ok = rok && l < L_eff;
        float t = fmaf(o[e], osc_r, bias_r);
        if (rb) t += ok ? rb[l] : 0.f;
        if (r2b) t = (ok ? r2b[l] : 0.f) + t;
        if (use_div) t = t / d.div;
        if (ok) {
          yb[l] = t;
          s1 += t;
          s2 = fmaf(t, t, s2);
        }
      }
    }
    asm volatile("" : "+v"(s1), "+v"(s2));
  });
  if (d.part) {
    const float a1 = s1 + __shfl_xor(s1, 32, 64);
    const float a2 = s2 + __shfl_xor(s2, 32, 64);
    const int pblk = blockIdx.x * WN + wn;  // one partial per wave block of 32 TN tiles
    if (kg == 0 && rok && pblk < d.part_nt) {
      float2* pp = reinterpret_cast<float2*>(d.part) + ((int64_t)b * d.C_out + co) * d.part_nt + pblk;
      *pp = make_float2(a1, a2);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// activation + input transform pass: x fp32 [B][C][L] -> vs (see WArgs).  Workgroup = (256 tiles, one group of 8 channels,
// one batch item): the 770 input positions it needs are activated ONCE into LDS (coalesced reads along l), then every
// thread transforms its tile's 5 inputs x 8 channels and writes 5 points x (hi, lo) 16-byte slots (contiguous per wave).
// ---------------------------------------------------------------------------------------------------------------------
struct AArgs {
  const float* x; int64_t x_bs; int x_cs;
  int C, L, pad;
  int pro;  // 0 = none, 1 = AdaIN + Snake (st2_actsplit.hip ST2_PRO_ADAIN_SNAKE)
  const float* stats; const float* gamma; const float* beta; int64_t gb_bs; const float* alpha;
  float x_scale;
  h8* vs; int cg_tot; int Lt;
  int dil;      // conv dilation d: grid z = B * d virtual batch items, item (b, r) transforms a_r[q] = a[d q + r]; pad in q units
  int src_dil;  // > 1: x is the residue-major output of a dilated layer, x[(b * src_dil + l % src_dil)][ci][l / src_dil]
  Toom tm;
};

template <class S>
struct ActGeom {
  static constexpr int TILES = S::M <= 4 ? 256 : 128;     // tiles per workgroup (the activated positions must fit in LDS)
  static constexpr int POS = S::M * TILES + S::R - 1;     // input positions per workgroup
  static constexpr int PITCH = POS | 1;                   // odd pitch: the 8 channel rows start in different banks
};

template <class S, int PRO>
__global__ __launch_bounds__(256) void act_w3_kernel(const AArgs a) {
  constexpr int P = S::P, M = S::M, AT_TILES = ActGeom<S>::TILES, AT_POS = ActGeom<S>::POS, AT_PITCH = ActGeom<S>::PITCH;
  __shared__ float sa[8 * AT_PITCH];
  const int tile0 = blockIdx.x * AT_TILES;
  const int cg = blockIdx.y;
  const int vb = blockIdx.z;
  const int b = vb / a.dil, r = vb - b * a.dil;
  const int p0 = M * tile0 - a.pad;  // first input position (in q units) of the workgroup
  for (int idx = threadIdx.x; idx < 8 * AT_POS; idx += 256) {
    const int e = idx / AT_POS, i = idx - e * AT_POS;
    const int q = p0 + i;
    const int l = a.dil * q + r;  // natural position
    const int ci = cg * 8 + e;
    float u = 0.f;
    if (q >= 0 && l < a.L && ci < a.C) {
      if (a.src_dil > 1) {
        const int rs = l % a.src_dil, qs = l / a.src_dil;
        u = a.x[((int64_t)b * a.src_dil + rs) * a.x_bs + (int64_t)ci * a.x_cs + qs];
      } else {
        u = a.x[(int64_t)b * a.x_bs + (int64_t)ci * a.x_cs + l];
      }
      if constexpr (PRO == 1) {
        const float* st = a.stats + ((int64_t)b * a.C + ci) * 2;
        const float g = 1.0f + a.gamma[(int64_t)b * a.gb_bs + ci];
        const float bt = a.beta[(int64_t)b * a.gb_bs + ci];
        float w = (u - st[0]) * st[1];
        w = g * w + bt;
        const float al = a.alpha[ci];
        u = snake(w, al, 1.0f / al);
      }
      u *= a.x_scale;
    }
    sa[e * AT_PITCH + i] = u;  // zero outside the tensor: F.conv1d pads the ACTIVATED tensor
  }
  __syncthreads();
  const int T = tile0 + threadIdx.x;
  if ((int)threadIdx.x >= AT_TILES || T >= a.Lt) return;
  h8 hi[P], lo[P];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float* s = sa + e * AT_PITCH + M * threadIdx.x;
    float dd[P];
#pragma unroll
    for (int n = 0; n < P; ++n) dd[n] = s[n];
#pragma unroll
    for (int p = 0; p < P; ++p) {
      float v = a.tm.BT[p][0] * dd[0];
#pragma unroll
      for (int n = 1; n < P; ++n) v = fmaf(a.tm.BT[p][n], dd[n], v);
      const float uc = st2_clamp_f16(v);
      const _Float16 h = (_Float16)uc;
      hi[p][e] = h;
      lo[p][e] = (_Float16)(uc - (float)h);
    }
  }
  const int64_t pstride = (int64_t)a.cg_tot * a.Lt;
  h8* dst = a.vs + (int64_t)vb * 2 * P * pstride + (int64_t)cg * a.Lt + T;
#pragma unroll
  for (int p = 0; p < P; ++p) {
    dst[p * pstride] = hi[p];
    dst[(P + p) * pstride] = lo[p];
  }
}

}  // namespace st2w
