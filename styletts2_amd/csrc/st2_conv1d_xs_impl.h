// Conv1d on the CDNA4 f16 matrix pipe over PRE-ACTIVATED, PRE-SPLIT operands (the "xs" path).
//
//   y[b,co,l] = epi( bias[co] + sum_{ci,t} W[co,ci,t] * a[b,ci, l + t*dil - pad_left] )
//
// `a` = x_scale * pro(x) arrives from st2_act_split (st2_actsplit.hip) as two f16 planes (hi, lo) laid out in
// 16-byte slots of 8 consecutive channels, [b][plane][ci/8][pos], with the conv's zero padding already in place
// (halo columns on the left, a zero tail on the right, zero channel padding).  The MFMA kernel therefore does no
// per-element arithmetic and tests no boundary: a chunk of CI_T channels is 2*CI_T/8 rows of XW = BN + (ks-1)*dil
// slots that are copied global -> registers -> LDS as 16-byte vectors, double buffered (loads for chunk c+1 are
// issued at the first k-step of chunk c, written to the other LDS buffer at its last k-step; one barrier per chunk).
//
// GEMM structure (as st2_conv1d_f16s.hip): per batch item M = C_out, N = L_out, K = C_in*ks ordered (ci/16, tap,
// ci%16); product = hi_w*hi_a + hi_w*lo_a + lo_w*hi_a on v_mfma_f32_32x32x16_f16 into one fp32 accumulator; the B
// fragment is one conflict-free ds_read_b128 per (tap, 32 columns), a tap is a shift of the slot index; the A
// fragments (weights, L2 resident, every wave owns distinct output rows) go straight from L2 to registers, one
// k-step ahead.  Wave tile 32 (co) x 128 (l); workgroup = 4 waves as 4x1 / 2x2 / 1x4 over (co, l) by C_out.
//
// Epilogue: out_scale, bias, residual(s), divide, activation, coalesced fp32 stores -- and, if d.part is given, the
// per-tile (sum, sum of squares) of the stored values per output channel (half-wave reduce-scatter over the 32
// column lanes, fixed order) so the next layer's InstanceNorm statistics cost no extra pass over the tensor.
#pragma once
#include "st2_common.h"
#include "st2_act.h"
#include <type_traits>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int NT = 256;

// Sum 4 per-lane values over the 32 lanes that share (lane >> 5).  Reduce-scatter: after the exchange rounds lane l
// holds the total of element 2*bit4(l) + bit3(l) (all 8 lanes of that group hold it).  6 cross-lane moves instead of
// 20; summation order is fixed (bitwise reproducible).  Done once per group of 4 output rows so that only 2 x 4
// partial sums are live next to the 64 accumulators (the 2 x 16 of a whole-tile reduction made the 168-VGPR build
// spill in its epilogue).
__device__ __forceinline__ float halfwave_reduce4(const float (&s)[4], int l31) {
  float a2[2];
  bool up = (l31 & 16) != 0;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const float mine = up ? s[2 + i] : s[i];
    const float theirs = up ? s[i] : s[2 + i];
    a2[i] = mine + __shfl_xor(theirs, 16, 64);
  }
  up = (l31 & 8) != 0;
  const float mine = up ? a2[1] : a2[0];
  const float theirs = up ? a2[0] : a2[1];
  float a1 = mine + __shfl_xor(theirs, 8, 64);
  a1 += __shfl_xor(a1, 4, 64);
  a1 += __shfl_xor(a1, 2, 64);
  a1 += __shfl_xor(a1, 1, 64);
  return a1;
}

// Two builds of the body: one held to 2 workgroups per CU (<= 256 registers; the variants with wide staging tiles) and
// one capped at 168 VGPRs (3 workgroups per CU: a third wave per SIMD to hide LDS / L2 latency behind; measured
// 0.42 ms vs 0.48 ms on the dominant layer at B = 8).
template <int KS, int CI_T, int WM, int WN, int TN>
__device__ __forceinline__ void conv1d_xs_body(const st2_conv_desc& d) {
  constexpr int BM = 32 * WM;
  constexpr int BN = 32 * TN * WN;
  constexpr int CG = CI_T / 8;     // 8-channel groups per chunk
  constexpr int ROWS = 2 * CG;     // staged rows per chunk: (plane, group)
  constexpr int S16 = CI_T / 16;   // MFMA k-steps per tap per chunk
  constexpr int MAXXW = BN + (KS - 1) * 8;
  constexpr int NS = (ROWS * MAXXW + NT - 1) / NT;  // staged 16-byte slots per thread per chunk
  static_assert(WM * WN == 4, "4 waves");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  h8* lds = reinterpret_cast<h8*>(smem_raw);  // [2 buffers][NS*NT >= ROWS*XW] slots of 16 B, image = [ROWS][XW]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int kg = lane >> 5;
  const int l31 = lane & 31;
  const int wm = wave / WN;
  const int wn = wave % WN;
  const int n0 = blockIdx.x * BN;
  const int m0 = blockIdx.y * BM;
  const int b = blockIdx.z;

  const int XW = BN + (KS - 1) * d.dil;  // staged positions per row
  const int S = ROWS * XW;               // staged slots per chunk
  const int Lp = d.xs_lp;
  const int64_t gplane = (int64_t)d.xs_cg * Lp;  // slots per plane of one batch item
  // slot (row, col) of the chunk image <- xs[b][plane = row / CG][c*CG + row % CG][n0 - pad_left + halo + col]
  const h8* xsb = reinterpret_cast<const h8*>(d.xs) + (int64_t)b * 2 * gplane + (n0 - d.pad_left + d.xs_halo);
  // Loads and LDS stores are UNCONDITIONAL (slots past the image re-read slot 0 and land in the buffer's slack):
  // with predicated loads hipcc cannot count the in-order VMEM queue and drains it at the next weight wait.
  int soff[NS];
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const int slot = tid + i * NT;
    const int row = slot / XW;
    const int col = slot - row * XW;
    soff[i] = slot < S ? (int)((row / CG) * gplane + (int64_t)(row % CG) * Lp + col) : 0;
  }
  constexpr int LBUF = NS * NT;  // LDS slots per buffer (>= S)
  h8 xr[NS];
  auto load_chunk = [&](int c) __attribute__((always_inline)) {
    const h8* src = xsb + (int64_t)c * CG * Lp;
#pragma unroll
    for (int i = 0; i < NS; ++i) xr[i] = src[soff[i]];
  };
  auto store_chunk = [&](int buf) __attribute__((always_inline)) {
    h8* dst = lds + (size_t)buf * LBUF;
#pragma unroll
    for (int i = 0; i < NS; ++i) dst[tid + i * NT] = xr[i];
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
  const int nchunk = d.wq_cin_pad / CI_T;

  load_chunk(0);
  constexpr int SPC = S16 * KS;  // k-steps per chunk
  // Weight fragments run TWO k-steps ahead in three NAMED register sets (set = k-step index within the chunk mod 3,
  // a compile-time constant after unrolling).  VMEM returns in order, so the first weight wait that also has to
  // drain the activation loads of the next chunk (issued at k-step 0, after that step's prefetch) is the one of
  // k-step 3: three k-steps (>= 1100 MFMA cycles per wave) of slack for their HBM latency.
  h8 a_hi[3], a_lo[3];
  a_hi[0] = ap[0];
  a_lo[0] = ap[1];
  const int nsteps = nchunk * SPC;
  if (nsteps > 1) ap += a_step;
  a_hi[1] = ap[0];
  a_lo[1] = ap[1];
  store_chunk(0);
  __syncthreads();

  const int plane = CG * XW;  // LDS slots per plane of a chunk image
  // Every load and LDS store below is issued unconditionally (the last chunk re-stages itself into the idle buffer
  // and re-reads its last weight fragment): a branch around VMEM makes hipcc's in-order vmcnt bookkeeping
  // conservative and the next weight wait then drains the activation loads at HBM latency.
  // Workgroups sharing a CU run out of phase: while this wave is in its k loop, a neighbour's may be in its epilogue
  // (VALU + global memory).  Raised priority for the k loop keeps the matrix pipe fed first (cdna_hip_programming.md
  // T5: pays where waves have different roles); dropped again before the epilogue.
  __builtin_amdgcn_s_setprio(1);
  for (int c = 0; c < nchunk; ++c) {
    const int buf = c & 1;
    const bool more = c + 1 < nchunk;
    const h8* xbuf = lds + (size_t)buf * LBUF + kg * XW + wn * (32 * TN) + l31;
#pragma unroll
    for (int s = 0; s < S16; ++s) {
#pragma unroll
      for (int t = 0; t < KS; ++t) {
        const int i = s * KS + t;           // k-step within the chunk (compile-time after unrolling)
        const int cur = i % 3, pre = (i + 2) % 3;
        if (more || i + 2 < SPC) ap += a_step;  // scalar select, no branch around the loads
        a_hi[pre] = ap[0];                      // prefetch the weights of k-step i + 2
        a_lo[pre] = ap[1];
        // next chunk's activations: issued AFTER this step's weight prefetch (see above)
        if (i == 0) load_chunk(more ? c + 1 : c);
        // ... and parked in the other LDS buffer at the chunk's last k-step (free since the previous barrier)
        if (i == SPC - 1) store_chunk(buf ^ 1);
        __builtin_amdgcn_sched_barrier(0x786);  // neither VMEM nor MFMA crosses: the prefetch distance is kept
        const h8 ah = a_hi[cur], al = a_lo[cur];
        const h8* xp = xbuf + (2 * s) * XW + t * d.dil;
        h8 bh[TN], bl[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          bh[j] = xp[j * 32];
          bl[j] = xp[plane + j * 32];
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[j], acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[j], acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[j], acc[j], 0, 0, 0);
      }
    }
    // the next chunk indexes its steps from 0 again: rotate the two live sets (steps SPC, SPC+1) to sets 0, 1
    if (SPC % 3 == 1) {
      const h8 th = a_hi[1], tl = a_lo[1];  // sets (1, 2) -> (0, 1)
      a_hi[1] = a_hi[2];
      a_lo[1] = a_lo[2];
      a_hi[0] = th;
      a_lo[0] = tl;
    } else if (SPC % 3 == 2) {
      const h8 th = a_hi[0], tl = a_lo[0];  // sets (2, 0) -> (0, 1)
      a_hi[0] = a_hi[2];
      a_lo[0] = a_lo[2];
      a_hi[1] = th;
      a_lo[1] = tl;
    }
    __syncthreads();
  }

  __builtin_amdgcn_s_setprio(0);
  // ---- epilogue ---------------------------------------------------------------------------------------
  float* yb = d.y + (int64_t)b * d.y_bs;
  const float* rb = d.res ? d.res + (int64_t)b * d.res_bs : nullptr;
  const float* r2b = d.res2 ? d.res2 + (int64_t)b * d.res2_bs : nullptr;
  const float osc = d.out_scale;
  // per-row weight scale: unconditional load + select (the packed weights serve as a valid address without one)
  const float* rsc = d.w_row_scale ? d.w_row_scale : reinterpret_cast<const float*>(d.wq);
  const bool want_part = d.part != nullptr;
  const int ptile = blockIdx.x * WN + wn;  // 128-column tile index of this wave's partial sums
  // The epilogue comes in straight-line builds.  Which terms exist (residual, MRF accumulator, divide) is uniform per
  // launch; tested per element it turns the loop into thousands of one-store basic blocks whose residual loads are
  // each waited for on the spot (measured: the epilogue then costs as much as the k loop).  So interior tiles -- every
  // tile but the last along l / co -- of the plain-output convs dispatch ONCE to a build with those terms as
  // compile-time constants: no bounds tests, one 64-bit address per output row (the four 32-column groups of a lane
  // are immediate offsets), the row's residual loads issued together ahead of the arithmetic.  Edge tiles and rare
  // combinations take the generic build (MODE < 0: run-time flags, per-element bounds).
  const int col0 = n0 + wn * (32 * TN) + l31;
  const bool full_tile = m0 + BM <= d.C_out && n0 + BN <= d.L_out;  // workgroup-uniform
  const int rstep = 32 >> d.res_shift;
  auto epilogue_as = [&](auto act_tag, auto mode_tag) __attribute__((always_inline)) {
    constexpr int ACT = decltype(act_tag)::value;
    constexpr int MODE = decltype(mode_tag)::value;  // < 0: generic; else bit 0 = res, bit 1 = res2, bit 2 = div
    constexpr bool FULL = MODE >= 0;
    const bool use_res = FULL ? (MODE & 1) != 0 : rb != nullptr;
    const bool use_res2 = FULL ? (MODE & 2) != 0 : r2b != nullptr;
    const bool use_div = FULL ? (MODE & 4) != 0 : d.div != 1.0f;
    float ps[4], pq[4];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
      const bool rok = FULL || row < d.C_out;
      const int rowc = FULL ? row : min(row, d.C_out - 1);
      // 32-bit element offsets from the (scalar) per-batch bases: one VALU mad per row and tensor, and the memory
      // instructions take the SGPR-base + VGPR-offset form (a batch item is < 2^31 elements, checked at launch)
      const int yo = rowc * d.y_cs + col0;
      const int ro = rowc * d.res_cs + (col0 >> d.res_shift);  // used only if use_res
      const int r2o = rowc * d.res2_cs + col0;                 // used only if use_res2
      // unconditional load + select (a branch here would split the rows into separate basic blocks); without a bias
      // the packed weights serve as a valid address
      const float braw = (d.bias ? d.bias : reinterpret_cast<const float*>(d.wq))[rowc];
      const float bias_r = d.bias ? braw : 0.f;
      const float sraw = rsc[row];  // row < wq_co_pad by construction of the packing
      const float osc_r = d.w_row_scale ? osc * sraw : osc;
      bool ok[TN];
      float rv[TN], r2v[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        ok[j] = FULL || (rok && col0 + j * 32 < d.L_out);
        rv[j] = (use_res && ok[j]) ? rb[ro + j * rstep] : 0.f;
        r2v[j] = (use_res2 && ok[j]) ? r2b[r2o + j * 32] : 0.f;
      }
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        float v = acc[j][r] * osc_r + bias_r;
        if (use_res) v += rv[j];
        if (use_res2) v = r2v[j] + v;
        if (use_div) v = v / d.div;
        if constexpr (ACT == ST2_ACT_GELU) {
          v = gelu_erf(v);
        } else if constexpr (ACT == ST2_ACT_EXP_SIN) {
          v = row < d.act_split ? expf(v) : sin_acc(v);
        } else if constexpr (ACT == ST2_ACT_TANH) {
          v = tanhf(v);
        } else if constexpr (ACT == ST2_ACT_LEAKY) {
          v = leaky(v, d.act_slope);
        } else if constexpr (ACT == ST2_ACT_GELU_TANH) {
          v = gelu_tanh(v);
        }
        if (ok[j]) {
          yb[yo + j * 32] = v;
          s1 += v;
          s2 += v * v;
        }
      }
      ps[r & 3] = s1;
      pq[r & 3] = s2;
      if ((r & 3) == 3) {
        if (want_part) {  // wave-uniform: the (sum, sumsq) of these 4 rows over the wave's 128 columns
          const float ts = halfwave_reduce4(ps, l31);
          const float tq = halfwave_reduce4(pq, l31);
          const int rr = (r & ~3) + ((l31 >> 4) & 1) * 2 + ((l31 >> 3) & 1);
          const int prow = m0 + wm * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * kg;
          if ((l31 & 7) == 0 && prow < d.C_out && ptile < d.part_nt) {
            float2* pp = reinterpret_cast<float2*>(d.part) + ((int64_t)b * d.C_out + prow) * d.part_nt + ptile;
            *pp = make_float2(ts, tq);
          }
        }
        __builtin_amdgcn_sched_barrier(0);  // four rows of loads in flight at a time (VGPR budget)
      }
    }
  };
  auto epilogue = [&](auto act_tag) __attribute__((always_inline)) {
    constexpr int ACT = decltype(act_tag)::value;
    const int mode = (rb ? 1 : 0) | (r2b ? 2 : 0) | (d.div != 1.0f ? 4 : 0);
    if (!full_tile) return epilogue_as(act_tag, std::integral_constant<int, -1>{});
    if constexpr (ACT == ST2_ACT_NONE) {
      switch (mode) {
        case 0: return epilogue_as(act_tag, std::integral_constant<int, 0>{});
        case 1: return epilogue_as(act_tag, std::integral_constant<int, 1>{});
        case 2: return epilogue_as(act_tag, std::integral_constant<int, 2>{});
        case 3: return epilogue_as(act_tag, std::integral_constant<int, 3>{});
        case 4: return epilogue_as(act_tag, std::integral_constant<int, 4>{});
        case 5: return epilogue_as(act_tag, std::integral_constant<int, 5>{});
        case 6: return epilogue_as(act_tag, std::integral_constant<int, 6>{});
        default: return epilogue_as(act_tag, std::integral_constant<int, 7>{});
      }
    } else {
      if (mode == 0) return epilogue_as(act_tag, std::integral_constant<int, 0>{});
      return epilogue_as(act_tag, std::integral_constant<int, -1>{});
    }
  };
  switch (d.act) {
    case ST2_ACT_GELU:
      epilogue(std::integral_constant<int, ST2_ACT_GELU>{});
      break;
    case ST2_ACT_EXP_SIN:
      epilogue(std::integral_constant<int, ST2_ACT_EXP_SIN>{});
      break;
    case ST2_ACT_TANH:
      epilogue(std::integral_constant<int, ST2_ACT_TANH>{});
      break;
    case ST2_ACT_LEAKY:
      epilogue(std::integral_constant<int, ST2_ACT_LEAKY>{});
      break;
    case ST2_ACT_GELU_TANH:
      epilogue(std::integral_constant<int, ST2_ACT_GELU_TANH>{});
      break;
    default:
      epilogue(std::integral_constant<int, ST2_ACT_NONE>{});
      break;
  }
}

template <int KS, int CI_T, int WM, int WN, int TN, int OCC>
__global__ __launch_bounds__(NT, 2) void conv1d_xs_kernel(const st2_conv_desc d) {  // >= 2 workgroups per CU
  conv1d_xs_body<KS, CI_T, WM, WN, TN>(d);
}
template <int KS, int CI_T, int WM, int WN, int TN>
__global__ __launch_bounds__(NT, 3) void conv1d_xs_kernel_o3(const st2_conv_desc d) {  // <= 168 VGPRs
  conv1d_xs_body<KS, CI_T, WM, WN, TN>(d);
}

template <int KS, int CI_T, int WM, int WN, int TN, int OCC>
int launch(const st2_conv_desc& d, hipStream_t s) {
  constexpr int BM = 32 * WM;
  constexpr int BN = 32 * TN * WN;
  const int XW = BN + (KS - 1) * d.dil;
  const int C_pad = (d.C_in + CI_T - 1) / CI_T * CI_T;
  constexpr int NS = ((2 * CI_T / 8) * (BN + (KS - 1) * 8) + NT - 1) / NT;
  const size_t smem = (size_t)2 * NS * NT * 16;
  ST2_REQUIRE(smem <= 160 * 1024, "st2_conv1d_xs: tile needs %zu B of LDS (ks=%d dil=%d)", smem, KS, d.dil);
  ST2_REQUIRE(d.wq_cin_pad == C_pad, "st2_conv1d_xs: packed weight has %d input channels, kernel needs %d",
              d.wq_cin_pad, C_pad);
  ST2_REQUIRE(d.wq_co_pad % BM == 0 && d.wq_co_pad >= d.C_out, "st2_conv1d_xs: wq_co_pad=%d must be a multiple "
              "of %d covering C_out=%d", d.wq_co_pad, BM, d.C_out);
  ST2_REQUIRE(d.xs_cg * 8 >= C_pad, "st2_conv1d_xs: xs has %d channel groups, kernel needs %d", d.xs_cg, C_pad / 8);
  const int n_tiles = st2_cdiv(d.L_out, BN);
  // last slot the last tile stages: (n_tiles-1)*BN - pad_left + halo + XW - 1
  ST2_REQUIRE((int64_t)(n_tiles - 1) * BN - d.pad_left + d.xs_halo + XW <= d.xs_lp,
              "st2_conv1d_xs: xs rows of %d slots are too short for L_out=%d (tile %d, ks=%d, dil=%d, halo=%d)",
              d.xs_lp, d.L_out, BN, KS, d.dil, d.xs_halo);
  if (d.part) ST2_REQUIRE(d.part_nt >= st2_cdiv(d.L_out, 128), "st2_conv1d_xs: part_nt=%d < %d tiles", d.part_nt,
                          st2_cdiv(d.L_out, 128));
  static bool attr_done = false;
  if (!attr_done) {
    if constexpr (OCC == 3)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1d_xs_kernel_o3<KS, CI_T, WM, WN, TN>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    else
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1d_xs_kernel<KS, CI_T, WM, WN, TN, 2>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  dim3 grid(n_tiles, st2_cdiv(d.C_out, BM), d.B);
  if constexpr (OCC == 3)
    hipLaunchKernelGGL((conv1d_xs_kernel_o3<KS, CI_T, WM, WN, TN>), grid, dim3(NT), smem, s, d);
  else
    hipLaunchKernelGGL((conv1d_xs_kernel<KS, CI_T, WM, WN, TN, 2>), grid, dim3(NT), smem, s, d);
  ST2_CHECK_LAUNCH("st2_conv1d_xs");
  return 0;
}

}  // namespace

namespace st2xs {

template <int KS, int CI_T>
int launch_by_cout(const st2_conv_desc& d, hipStream_t s) {
  if (d.C_out > 64) return launch<KS, CI_T, 4, 1, 4, 3>(d, s);  // 128 co x 128 l, 3 workgroups / CU
  if (d.C_out > 32) {                                           // 64 co x 256 l
    if constexpr (CI_T == 16)
      return launch<KS, CI_T, 2, 2, 4, 3>(d, s);
    else
      return launch<KS, CI_T, 2, 2, 4, 2>(d, s);  // the 168-VGPR build spills with 32-channel chunks
  }
  return launch<KS, CI_T, 1, 4, 4, 2>(d, s);  // 32 co x 512 l
}

}  // namespace st2xs
