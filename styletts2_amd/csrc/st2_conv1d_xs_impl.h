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
// GEMM structure: per batch item M = C_out, N = L_out, K = C_in*ks ordered (ci/16, tap, ci%16); product =
// hi_w*hi_a + hi_w*lo_a + lo_w*hi_a on v_mfma_f32_32x32x16_f16 into one fp32 accumulator; the activation fragment is
// one conflict-free ds_read_b128 per (tap, 32 columns), a tap is a shift of the slot index; the weight fragments (L2
// resident, every wave owns distinct output rows) go straight from L2 to registers, two k-steps ahead.  The
// activations are the MFMA's A operand and the weights its B operand, so the accumulator is the TRANSPOSED tile: a
// lane owns one output row and runs of 4 consecutive positions.  Wave tile 32 (co) x 128 (l); workgroup = 4 waves as
// 4x1 / 2x2 / 1x4 over (co, l) by C_out.
//
// Epilogue: out_scale, bias, residual(s), divide, activation, 16-byte fp32 stores / residual loads per lane on interior
// tiles of aligned tensors -- and, if d.part is given, the per-tile (sum, sum of squares) of the stored values per
// output channel (a lane's own values + one cross-lane move, fixed order) so the next layer's InstanceNorm statistics
// cost no extra pass over the tensor.
#pragma once
#include "st2_conv_epilogue.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef st2_f32x16 f32x16;

// Instrumentation switches of the micro-benchmark tools/xs_bench.hip (all compiled out of the library, ST2_XS_ABLATE = 0):
// bit 0 = activation fragments read from LDS once per chunk instead of per k-step, bit 1 = weight fragments loaded once,
// bit 2 = no epilogue, bit 3 = activations staged once (what does each stream cost on top of the bare MFMA loop?), bit 6
// = per-workgroup timeline (s_memtime / s_memrealtime at start, k-loop end and exit, HW_ID, XCC_ID) into d.stats.
#ifndef ST2_XS_ABLATE
#define ST2_XS_ABLATE 0
#endif
#ifndef ST2_XS_NSET
#define ST2_XS_NSET 3  // weight-fragment register sets = prefetch distance + 1
#endif
// Bisection switches of tools/lstm_load_repro.hip (round 6: which property of this kernel disturbs another queue's loads?):
// ST2_XS_SETPRIO = 0 drops the raised wave priority of the k loop; ST2_XS_PRED_STAGE = 1 predicates the staging loads / LDS
// stores of slots past the chunk image instead of re-reading slot 0 (narrow tiles: up to 70 % of the lanes).
#ifndef ST2_XS_SETPRIO
#define ST2_XS_SETPRIO 1
#endif
#ifndef ST2_XS_PRED_STAGE
#define ST2_XS_PRED_STAGE 0
#endif

namespace {

constexpr int NT = 256;
constexpr int ABL = ST2_XS_ABLATE;

// Two builds of the body: one held to 2 workgroups per CU (<= 256 registers; the variants with wide staging tiles) and
// one capped at 168 VGPRs (3 workgroups per CU: a third wave per SIMD to hide LDS / L2 latency behind; measured
// 0.42 ms vs 0.48 ms on the dominant layer at B = 8).
// (n0, by, bz) = the tile's (first column, row block, batch item): the launch wrappers below derive it from blockIdx, with
// or without the XCD-aware remap.  TN need not be the launch's: a tile at the end of a row may run a narrower body.
template <int KS, int CI_T, int WM, int WN, int TN>
__device__ __forceinline__ void conv1d_xs_body(const st2_conv_desc& d, const int n0, const unsigned by, const unsigned bz,
                                               const int tid) {
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

  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int kg = lane >> 5;
  const int l31 = lane & 31;
  const int wm = wave / WN;
  const int wn = wave % WN;
  const int m0 = by * BM;
  const int b = bz;

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
    for (int i = 0; i < NS; ++i) {
      if constexpr (ST2_XS_PRED_STAGE) {
        if (tid + i * NT < S) xr[i] = src[soff[i]];
      } else {
        xr[i] = src[soff[i]];
      }
    }
  };
  auto store_chunk = [&](int buf) __attribute__((always_inline)) {
    h8* dst = lds + (size_t)buf * LBUF;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      if constexpr (ST2_XS_PRED_STAGE) {
        if (tid + i * NT < S) dst[tid + i * NT] = xr[i];
      } else {
        dst[tid + i * NT] = xr[i];
      }
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
  const int nchunk = d.wq_cin_pad / CI_T;

  constexpr int SPC = S16 * KS;  // k-steps per chunk
  unsigned long long tl_t0 = 0, tl_r0 = 0;
  if constexpr (ABL & 64) {
    tl_t0 = __builtin_amdgcn_s_memtime();
    tl_r0 = __builtin_amdgcn_s_memrealtime();
  }
  load_chunk(0);
  // Weight fragments run NSET - 1 k-steps ahead in NSET NAMED register sets (set = k-step index within the chunk mod
  // NSET, a compile-time constant after unrolling).  VMEM returns in order, so the first weight wait that also has to
  // drain the activation loads of the next chunk (issued at k-step 0, after that step's prefetch) is the one of
  // k-step NSET: NSET k-steps (>= 380 MFMA cycles each per wave) of slack for their HBM latency.  The distance has to
  // cover an L2 round trip at the pace of a wave that runs with ONE competitor on its SIMD (the third workgroup of
  // the CU being in its epilogue most of the time): per-workgroup s_memtime stamps showed the matrix pipe idle 23 %
  // of the cycles with two k-steps of distance.
  constexpr int NSET = ST2_XS_NSET;
  h8 a_hi[NSET], a_lo[NSET];
  const int nsteps = nchunk * SPC;
#pragma unroll
  for (int k = 0; k < NSET - 1; ++k) {
    if (k > 0 && k < nsteps) ap += a_step;
    a_hi[k] = ap[0];
    a_lo[k] = ap[1];
  }
  store_chunk(0);
  __syncthreads();

  const int plane = CG * XW;  // LDS slots per plane of a chunk image
  // Every load and LDS store below is issued unconditionally (the last chunk re-stages itself into the idle buffer
  // and re-reads its last weight fragment): a branch around VMEM makes hipcc's in-order vmcnt bookkeeping
  // conservative and the next weight wait then drains the activation loads at HBM latency.
  // Workgroups sharing a CU run out of phase: while this wave is in its k loop, a neighbour's may be in its epilogue
  // (VALU + global memory).  Raised priority for the k loop keeps the matrix pipe fed first (cdna_hip_programming.md
  // T5: pays where waves have different roles); dropped again before the epilogue.
  h8 abl_bh[TN], abl_bl[TN];  // ABL & 1 only
  if constexpr (ST2_XS_SETPRIO) __builtin_amdgcn_s_setprio(1);
  for (int c = 0; c < nchunk; ++c) {
    const int buf = c & 1;
    const bool more = c + 1 < nchunk;
    const h8* xbuf = lds + (size_t)buf * LBUF + kg * XW + wn * (32 * TN) + l31;
#pragma unroll
    for (int s = 0; s < S16; ++s) {
#pragma unroll
      for (int t = 0; t < KS; ++t) {
        const int i = s * KS + t;           // k-step within the chunk (compile-time after unrolling)
        const int cur = i % NSET, pre = (i + NSET - 1) % NSET;
        if constexpr (!(ABL & 2)) {
          if (more || i + NSET - 1 < SPC) ap += a_step;  // scalar select, no branch around the loads
          a_hi[pre] = ap[0];                             // prefetch the weights of k-step i + NSET - 1
          a_lo[pre] = ap[1];
        } else {
          a_hi[pre] = a_hi[cur];
          a_lo[pre] = a_lo[cur];
        }
        if constexpr (!(ABL & 8)) {
          // next chunk's activations: issued AFTER this step's weight prefetch (see above)
          if (i == 0) load_chunk(more ? c + 1 : c);
          // ... and parked in the other LDS buffer at the chunk's last k-step (free since the previous barrier)
          if (i == SPC - 1) store_chunk(buf ^ 1);
        }
        __builtin_amdgcn_sched_barrier(0x786);  // neither VMEM nor MFMA crosses: the prefetch distance is kept
        const h8 ah = a_hi[cur], al = a_lo[cur];
        const h8* xp = xbuf + (2 * s) * XW + t * d.dil;
        h8 bh[TN], bl[TN];
        if constexpr (ABL & 1) {
          if (i == 0) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              abl_bh[j] = xp[j * 32];
              abl_bl[j] = xp[plane + j * 32];
            }
          }
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            bh[j] = abl_bh[j];
            bl[j] = abl_bl[j];
          }
        } else {
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            bh[j] = xp[j * 32];
            bl[j] = xp[plane + j * 32];
          }
        }
#pragma unroll
        // activations as the MFMA's A operand, weights as B: the accumulator comes out TRANSPOSED -- lane = output row
        // (co), registers = positions, 4 consecutive l per (r >> 2) -- so the epilogue moves 16 bytes per lane and
        // instruction (see below); the products and their summation order are the same as with (weights, activations)
        for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], ah, acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[j], ah, acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], al, acc[j], 0, 0, 0);
      }
    }
    // the next chunk indexes its steps from 0 again: rotate the live sets (steps SPC .. SPC + NSET - 2) to 0 .. NSET - 2
    if constexpr (SPC % NSET != 0) {
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

  if constexpr (ST2_XS_SETPRIO) __builtin_amdgcn_s_setprio(0);
  unsigned long long tl_t1 = 0;
  if constexpr (ABL & 64) tl_t1 = __builtin_amdgcn_s_memtime();
  auto tl_write = [&]() __attribute__((always_inline)) {
   if constexpr (ABL & 64) {  // micro-benchmark timeline: d.stats (unused by this kernel) carries a uint64 buffer
    __builtin_amdgcn_s_waitcnt(0);  // stores issued (vmcnt / lgkmcnt drained as far as this counter encodes)
    if (tid == 0) {
      unsigned long long* tl = reinterpret_cast<unsigned long long*>(const_cast<float*>(d.stats));
      const unsigned long long lin = blockIdx.x + gridDim.x * (blockIdx.y + (unsigned long long)gridDim.y * blockIdx.z);
      tl[lin * 8 + 0] = (unsigned long long)__builtin_amdgcn_s_getreg(0xF804) |
                        ((unsigned long long)__builtin_amdgcn_s_getreg(0xF814) << 32);
      tl[lin * 8 + 1] = tl_t0;
      tl[lin * 8 + 2] = tl_t1;
      tl[lin * 8 + 3] = __builtin_amdgcn_s_memtime();
      tl[lin * 8 + 4] = tl_r0;
      tl[lin * 8 + 5] = __builtin_amdgcn_s_memrealtime();
    }
  }
  };
  if constexpr (ABL & 4) {  // keep the accumulators live, store (practically) nothing
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) t += acc[j][r];
    if (t == 12345.678f) d.y[tid] = t;
    tl_write();
    return;
  }
  // ---- epilogue (st2_conv_epilogue.h: shared with the fused kernel) --------------------------------------
  st2_conv_epilogue<TN, WM, WN>(d, acc, b, m0, n0, wm, wn, l31, kg);
  tl_write();
}

// Tile of workgroup / queue position `lin` in a grid of nx x ny x nz tiles.  `flags` bit 0 = XCD-aware order (see
// st2xs::XS_V_SWIZZLE): workgroup `lin` of a launch runs on XCD lin % 8 (observed dispatch order, MI355X_MICROARCH.md), and a
// launch whose output rows span ny = 2 / 4 / 8 row blocks gives every XCD ONE of them, so that an XCD's 4 MB L2 holds 1 / ny of
// the packed weights instead of all of them.  Scalar arithmetic only; the launcher checked ny and 8 | tile count.
__device__ __forceinline__ void xs_tile_of(unsigned lin, unsigned nx, unsigned ny, int flags, unsigned& bx, unsigned& by,
                                           unsigned& bz) {
  if (flags & 1) {
    const unsigned xcd = lin & 7, slot = lin >> 3;
    by = xcd % ny;
    const unsigned r = xcd / ny + (8 / ny) * slot;  // enumerates (l tile, batch item) within this row block
    bx = r % nx;
    bz = r / nx;
  } else {
    bx = lin % nx;
    const unsigned r = lin / nx;
    by = r % ny;
    bz = r / ny;
  }
}

// Row ends: a tile whose valid columns fit a quarter of the tile width (L = 400 in 128-column tiles: 16 columns; L = 8 000
// in 256-column tiles: 64) runs the body with a quarter of the column blocks -- the same k loop over the same staged rows,
// the same products in the same order for the columns that exist, and (the tile starts a 128-column group of its own) the
// same partial sums -- instead of multiplying three quarters of a tile of zero padding: 1 in 4 tiles at L = 400, 1 in 7 at
// L = 800, 1 in 32 on the first vocoder stage.
#ifndef ST2_XS_ROWEND
#define ST2_XS_ROWEND 1
#endif
#define ST2_XS_ONE_TILE_KERNEL(NAME, WGS_PER_CU)                                                                        \
  template <int KS, int CI_T, int WM, int WN, int TN>                                                                   \
  __global__ __launch_bounds__(NT, WGS_PER_CU) void NAME(const st2_conv_desc d, const int flags) {                     \
    unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;                                                         \
    if (flags & 1) xs_tile_of(bx + gridDim.x * (by + gridDim.y * bz), gridDim.x, gridDim.y, flags, bx, by, bz);         \
    const int n0 = (int)bx * (32 * TN * WN);                                                                            \
    if constexpr (ST2_XS_ROWEND && WN == 1 && TN >= 4) {                                                                \
      if (d.L_out - n0 <= 32 * (TN / 4)) { /* workgroup-uniform */                                                      \
        conv1d_xs_body<KS, CI_T, WM, WN, TN / 4>(d, n0, by, bz, threadIdx.x);                                           \
        return;                                                                                                         \
      }                                                                                                                 \
    }                                                                                                                   \
    conv1d_xs_body<KS, CI_T, WM, WN, TN>(d, n0, by, bz, threadIdx.x);                                                   \
  }
ST2_XS_ONE_TILE_KERNEL(conv1d_xs_kernel_o2, 2)  // >= 2 workgroups per CU (the variants with wide staging tiles)
ST2_XS_ONE_TILE_KERNEL(conv1d_xs_kernel_o3, 3)  // <= 168 VGPRs
ST2_XS_ONE_TILE_KERNEL(conv1d_xs_kernel_o4, 4)  // <= 128 VGPRs: narrow tiles only

template <int KS, int CI_T, int WM, int WN, int TN, int OCC>
int launch(const st2_conv_desc& d, hipStream_t s, bool swizzle = false) {
  constexpr int BM = 32 * WM;
  constexpr int BN = 32 * TN * WN;
  const int XW = BN + (KS - 1) * d.dil;
  // The packing's k order is (ci / 16, tap, ci % 16) whatever the chunk depth: any padding to a multiple of the chunk serves
  // (extra chunks are all-zero weights on the planes' zero channel padding).
  const int C_pad = d.wq_cin_pad;
  constexpr int NS = ((2 * CI_T / 8) * (BN + (KS - 1) * 8) + NT - 1) / NT;
  size_t smem = (size_t)2 * NS * NT * 16;
#ifdef ST2_XS_SMALL_LDS_PAD
  if (TN < 4) smem = std::max<size_t>(smem, (size_t)ST2_XS_SMALL_LDS_PAD * 1024);  // A/B: cap the narrow builds' workgroups per CU
#endif
  ST2_REQUIRE(smem <= 160 * 1024, "st2_conv1d_xs: tile needs %zu B of LDS (ks=%d dil=%d)", smem, KS, d.dil);
  ST2_REQUIRE(C_pad % CI_T == 0 && C_pad >= d.C_in && C_pad < d.C_in + 32,
              "st2_conv1d_xs: packed weight has %d input channels, kernel needs C_in=%d padded to a multiple of %d",
              d.wq_cin_pad, d.C_in, CI_T);
  ST2_REQUIRE(d.wq_co_pad % BM == 0 && d.wq_co_pad >= d.C_out, "st2_conv1d_xs: wq_co_pad=%d must be a multiple "
              "of %d covering C_out=%d", d.wq_co_pad, BM, d.C_out);
  ST2_REQUIRE(d.xs_cg * 8 >= C_pad, "st2_conv1d_xs: xs has %d channel groups, kernel needs %d", d.xs_cg, C_pad / 8);
  const int n_tiles = st2_cdiv(d.L_out, BN);
  // last slot the last tile stages: (n_tiles-1)*BN - pad_left + halo + XW - 1
  ST2_REQUIRE((int64_t)(n_tiles - 1) * BN - d.pad_left + d.xs_halo + XW <= d.xs_lp,
              "st2_conv1d_xs: xs rows of %d slots are too short for L_out=%d (tile %d, ks=%d, dil=%d, halo=%d)",
              d.xs_lp, d.L_out, BN, KS, d.dil, d.xs_halo);
  if constexpr (TN >= 4) {
    if (d.part)
      ST2_REQUIRE((d.part_cols == 0 || d.part_cols == 128) && d.part_nt >= st2_cdiv(d.L_out, 128),
                  "st2_conv1d_xs: part_nt=%d / part_cols=%d for %d tiles of 128 columns", d.part_nt, d.part_cols, st2_cdiv(d.L_out, 128));
  } else if (d.part) {  // one slot per tile (WN = 1): the caller sized `part` from st2_conv1d_xs_part_cols
    ST2_REQUIRE(WN == 1 && d.part_cols == BN && d.part_nt >= st2_cdiv(d.L_out, BN),
                "st2_conv1d_xs: a %d-column tile build needs part_cols=%d and part_nt >= %d (got %d / %d)", BN, BN,
                st2_cdiv(d.L_out, BN), d.part_cols, d.part_nt);
  }
  dim3 grid(n_tiles, st2_cdiv(d.C_out, BM), d.B);
  const int64_t total = (int64_t)grid.x * grid.y * grid.z;
  // XCD-aware order only where it is a bijection: 2 / 4 / 8 row blocks and a tile count divisible by 8
  const int flags = swizzle && (grid.y == 2 || grid.y == 4 || grid.y == 8) && total % 8 == 0;
  static std::atomic<uint64_t> attr_done{0};  // one bit per device ordinal (hipFuncSetAttribute is per device)
  st2_once_per_device(attr_done, [&] {
    if constexpr (OCC == 4)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1d_xs_kernel_o4<KS, CI_T, WM, WN, TN>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    else if constexpr (OCC == 3)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1d_xs_kernel_o3<KS, CI_T, WM, WN, TN>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    else
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1d_xs_kernel_o2<KS, CI_T, WM, WN, TN>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  if constexpr (OCC == 4)
    hipLaunchKernelGGL((conv1d_xs_kernel_o4<KS, CI_T, WM, WN, TN>), grid, dim3(NT), smem, s, d, flags);
  else if constexpr (OCC == 3)
    hipLaunchKernelGGL((conv1d_xs_kernel_o3<KS, CI_T, WM, WN, TN>), grid, dim3(NT), smem, s, d, flags);
  else
    hipLaunchKernelGGL((conv1d_xs_kernel_o2<KS, CI_T, WM, WN, TN>), grid, dim3(NT), smem, s, d, flags);
  ST2_CHECK_LAUNCH("st2_conv1d_xs");
  return 0;
}

}  // namespace

namespace st2xs {

// Builds of one launch ("variants": same products in the same order, same epilogue, bitwise the same results --
// tests/test_ops_gpu.py::test_xs_variants_are_bitwise_identical).  Which one is fastest depends on the box as well as on
// the shape: driver-class MI355X boxes run the 128 x 256 tile build 1.5-1.75 x slower than builder-class boxes on the
// C = 256 / L = 8 000 layers and only there, so the choice is MEASURED per shape class at start-up
// (st2_conv_tune, st2_conv1d_xs.hip) and falls back to the rule below when a class was not tuned.
//   bit 0  XS_V_WIDE     128 (co) x 256 (l) tiles, 2 workgroups / CU (k = 3, 7, 11) instead of 128 x 128, 3 workgroups / CU
//   bit 1  XS_V_SWIZZLE  XCD-aware tile order: one row block per XCD (launches with 2 / 4 / 8 row blocks)
// Tried as further variants in round 4 and removed again (bitwise equivalent, never a winner by the tuner's 2 % margin on any
// box: profiles/LAB_NOTES.md): 16-channel chunks at k = 3, and persistent workgroups pulling tiles from an atomic queue.
//   bit 2  XS_V_N64      128 (co) x 64 (l) tiles   } small grids (one utterance: long-form synthesis, B = 1 latency): a launch of fewer
//   bit 3  XS_V_N32      128 (co) x 32 (l) tiles   } workgroups than CUs is paced by ONE tile's k loop, so narrower tiles (2 x / 4 x the
//                        workgroups, each with 1/2 / 1/4 of the MFMAs per k-step) shorten it; y is bitwise the same (same products, same
//                        order per output element), the InstanceNorm partial sums come per 64 / 32 columns (d.part_cols) instead of 128
enum { XS_V_RULE = -1, XS_V_WIDE = 1, XS_V_SWIZZLE = 2, XS_V_N64 = 4, XS_V_N32 = 8 };

// The build an UNTUNED process runs (tests, one-off calls; a serving process measures: st2_conv_tune).  k >= 7: 32 (co) x
// 256 (l) wave tiles, 128 accumulator registers, 2 workgroups / CU -- half the weight stream (L2 -> registers) per FLOP;
// measured 1.59 vs 1.65 ms (k = 11) and 1.19 vs 1.23 ms (k = 7) at C = 128, L = 48 001, B = 32 (tools/xs_bench.hip,
// profiles/archive/r02/r02i_xs_bench_tn8.log) -- when the launch still has >= 2 rounds of workgroups at that tile size (512 slots): a
// single utterance (long-form synthesis, B = 1) keeps the 128-column tiles, which fill twice as many CUs.  Launches with
// 2 / 4 / 8 output row blocks (C_out = 256 ... 1024: the first vocoder stage) add the XCD-aware order: every XCD keeps one
// row block's weights in its L2 (k = 7, C = 256, L = 8 000: 0.570 vs 0.598 ms; k = 11: 0.851 vs 0.848, profiles/r04/r04q_bench.json).
// (Between rounds 2 and 4 the wide tiles were a trap on that stage: 32 tiles per row put every row-end tile on XCD 7, where
// the then whole-tile generic epilogue ran 3.5-12 x slower -- the "slow box class", DESIGN.md section 6 -- fixed in
// st2_conv_epilogue.h, which treats row ends by column blocks.)
inline int rule_variant(const st2_conv_desc& d) {
  if (d.C_out <= 64 || (d.ks < 7 && d.ks != 3)) return 0;
  const int ny = st2_cdiv(d.C_out, 128);
  if (d.ks == 3)  // long rows only: 0.400 -> 0.372 ms at C = 256, L = 8 000, -4 % at L = 800, +13 ... +70 % at L = 400 (r04ae)
    return (d.L_out >= 4096 && (int64_t)st2_cdiv(d.L_out, 256) * ny * d.B >= 1024) ? XS_V_WIDE : 0;
  const bool wide = (int64_t)st2_cdiv(d.L_out, 256) * ny * d.B >= 1024;
  const int64_t tiles = (int64_t)st2_cdiv(d.L_out, wide ? 256 : 128) * ny * d.B;
  const bool swz = (ny == 2 || ny == 4 || ny == 8) && tiles % 8 == 0;
  return (wide ? XS_V_WIDE : 0) | (swz ? XS_V_SWIZZLE : 0);
}

// Small grids.  A launch of a few hundred 128 x 128 tiles (one to three utterances: the long-form loop, B = 1 latency) does
// not fill 256 CUs x 3 workgroup slots; it is paced by ONE workgroup's k loop.  Halving / quartering the tile width doubles /
// quadruples the workgroups that share the weight stream of a k-step, so the loop gets shorter until the chip is full.
// Measured over 50 shapes at B = 1 ... 3 (tools/xs_bench.hip, profiles/r05/r05a_smallgrid_*, r05b_smallgrid_*): 32-column tiles
// win below ~100 tiles of 128 (k = 7, C = 256, L = 5 680, B = 1: 55.6 -> 28.3 us; k = 3: 39.6 -> 17.5), 64-column tiles up to ~600
// (k = 11, C = 128, L = 37 200: 75.3 -> 55.5 us; k = 3 up to ~900: 47.1 -> 27.4), 128 beyond -- for launches of up to THREE
// utterances: a batch of 8-32 short rows with the same tile count (k = 3, C = 256 ... 1024, L = 400 / 800, B = 32: the decoder
// front of the throughput configurations) LOSES 5-20 % with the narrow tiles (every workgroup streams its row block's whole
// weight slice, 0.4-1.6 MB there; profiles/r05/r05g_smallgrid_b32.log), so those keep the 128-column build.  A function of the geometry alone
// (never tuned: the partial sums' slot width follows it, and a measured choice would make the statistics box-dependent in
// their last bits).  Inside the pipeline (round 6, per-class times of the long-form and B = 1 legs with and without the k = 7 / 11
// narrow builds, profiles/r06/r06A_*): x1.25-1.46 at 90-291 tiles of 128 (k = 7, C = 128, L = 34 800: 95 -> 72 us; k = 11: 130 -> 100;
// k = 7, C = 256, L = 5 680: 77 -> 53), x0.95 at 376 (k = 11, L = 48 001: 64 -> 67.5 us) -- hence 340 rather than the micro-benchmark's
// 600 for k >= 7.  Callers opt in through d.part_cols (with statistics) or get it by rule (without): y is bitwise the same
// in every build.
#ifndef ST2_XS_NARROW_ALL
#define ST2_XS_NARROW_ALL 0  // tools/xs_bench.hip and the A/B probes build with 1: narrow tiles for every kernel size
#endif
#ifndef ST2_XS_SMALLGRID
#define ST2_XS_SMALLGRID 1  // build-time switch for A/B runs: 0 = every launch keeps the 128-column tiles
#endif
//
// k = 3 / 7 / 11.  (Round 5 held the k = 7 / 11 narrow builds back: with them on one queue the BiLSTM kernels on another returned
// different bits in 25-90 % of their calls.  Round 6 found the cause on the VICTIM's side -- a packed-f32 encoding gfx950 gets
// wrong next to the MFMA cadence of a one-accumulator tile, DESIGN.md section 9 -- removed it from the library and gates the
// build on its absence (tools/check_isa.py); tests/test_zz_coresidency_gpu.py runs every kernel class beside these builds.)
inline int small_grid_cols(const st2_conv_desc& d) {
  if (!ST2_XS_SMALLGRID || d.C_out <= 64 || (d.ks != 3 && d.ks != 7 && d.ks != 11) || d.B > 3) return 128;
  const int64_t wg128 = (int64_t)st2_cdiv(d.L_out, 128) * st2_cdiv(d.C_out, 128) * d.B;
  if (wg128 < 100) return 32;
  return wg128 < (d.ks == 3 ? 900 : 340) ? 64 : 128;
}

template <int KS, int CI_T>
int launch_by_cout(const st2_conv_desc& d, hipStream_t s, int variant) {
  if (variant < 0) variant = rule_variant(d);
  if (!(variant & (XS_V_N64 | XS_V_N32))) {  // (the micro-benchmark forces the bits; the library goes by the geometry)
    const int cols = d.part ? (d.part_cols ? d.part_cols : 128) : small_grid_cols(d);
    if (cols < 128) variant = (variant & XS_V_SWIZZLE) | (cols == 64 ? XS_V_N64 : XS_V_N32);
  }
  const bool swz = (variant & XS_V_SWIZZLE) != 0;
  if (d.C_out > 64) {
    if constexpr (KS == 3 || KS == 7 || KS == 11 || ST2_XS_NARROW_ALL) {  // small grids (see small_grid_cols)
      if (variant & XS_V_N32) return launch<KS, CI_T, 4, 1, 1, 3>(d, s, swz);
      if (variant & XS_V_N64) return launch<KS, CI_T, 4, 1, 2, 3>(d, s, swz);
    }
    if constexpr (KS >= 7 || KS == 3) {
      if (variant & XS_V_WIDE) return launch<KS, CI_T, 4, 1, 8, 2>(d, s, swz);
    }
    if constexpr (KS == 1 && CI_T == 32) {
      // Token GEMMs (the denoiser's / PL-BERT's Linears over the B*N merged tokens: C_out 512..1024 x 3 200 columns): at
      // 128 x 128 they are < 256 workgroups -- fewer than CUs, one per CU, nothing to hide the staging latency of a
      // 768-cycle chunk behind.  128 (co) x 64 (l) tiles double the workgroup count and 64-channel chunks double the
      // work between barriers: 1024 x 1024: 35.8 -> 28.9 us, 512 x 1024: 32.7 -> 17.7, 1024 x 2048: 63.6 -> 50.4, 768 x 768:
      // 27.9 -> 22.4 (profiles/experiments/gemm_bench.hip, profiles/archive/r03/r03c_gemm_bench.log); launches that already have >= 256 tiles
      // (C_out >= 2048) are fastest as they are.  Same products in the same order: results are bitwise unchanged.
      if ((int64_t)st2_cdiv(d.L_out, 128) * st2_cdiv(d.C_out, 128) * d.B < 256 && d.wq_cin_pad % 64 == 0 && !d.part)
        return launch<1, 64, 4, 1, 2, 3>(d, s, swz);
    }
    return launch<KS, CI_T, 4, 1, 4, 3>(d, s, swz);  // 128 co x 128 l, 3 workgroups / CU
  }
  if (d.C_out > 32) {                                           // 64 co x 256 l
    if constexpr (CI_T == 16)
      return launch<KS, CI_T, 2, 2, 4, 3>(d, s, swz);
    else
      return launch<KS, CI_T, 2, 2, 4, 2>(d, s, swz);  // the 168-VGPR build spills with 32-channel chunks
  }
  return launch<KS, CI_T, 1, 4, 4, 2>(d, s, swz);  // 32 co x 512 l
}

}  // namespace st2xs
