// Persistent stream-K build of the k = 1 xs conv -- the "token GEMMs": the denoiser's / PL-BERT's Linears as Conv1d(k = 1)
// over the B*N merged tokens (Modules/diffusion/modules.py:256-261, 484-490; Utils/PLBERT/util.py:6-12), M = C_out in
// 512 .. 2304, K = C_in in 512 .. 2048, N = 3 200 columns at B = 32.
//
// Why another schedule: as 128 x 128 / 128 x 64 tiles these launches are 200-450 workgroups for 256 CUs -- 1.3-1.6 rounds, a
// quarter of the CU-time idle -- and a 128-row tile re-reads its operands from L2 at ~60 B / clk / CU, the L2 -> CU rate, so
// 0.20-0.33 of the matrix roof was all they reached (profiles/archive/r03/r03c_gemm_bench.log, r03l_gemm_ablate.log).  Here:
//   * ONE persistent workgroup of 8 waves per CU computes 256 (co) x 128 (token) tiles: half the operand bytes per FLOP (the
//     token chunk staged in LDS feeds 8 row blocks instead of 4; every wave streams its own 32 weight rows L2 -> registers);
//   * the (tile, K-chunk) units of the launch are dealt out EVENLY: worker w takes units [U*w/W, U*(w+1)/W) in (tile,
//     chunk) order, so a worker's range is the tail of one tile, whole tiles, and the head of another (stream-K);
//   * a tile cut across workers meets in a fixed-order fix-up: whoever holds a tile's LAST chunk finishes it -- adds the
//     raw partial accumulators of the workers before it (decreasing worker index) and runs the shared epilogue.  Every
//     worker computes its head segment FIRST and publishes it (write-through 16-byte stores, then a flag), its finishing
//     segments afterwards: the partial a finisher needs was produced at the START of its neighbour's life, so it rarely
//     waits.  The split is a function of (geometry, worker count) alone: run-to-run bitwise reproducible; the summation
//     order differs from the one-tile-per-workgroup builds (same products, fp32 round-off class).
// Cross-CU hand-off follows MI355X_MICROARCH.md (workgroup dispatch & inter-workgroup visibility): producer = sc1 stores ->
// s_waitcnt vmcnt(0) -> barrier -> relaxed agent-scope flag store; consumer = relaxed poll -> agent acquire -> barrier ->
// plain loads; every spin is bounded (ST2_STATUS_GEMM_TIMEOUT).  A finisher only ever waits for LOWER-numbered workers.
// MEASURED AND NOT ADOPTED (round 4, profiles/r04/r04d_cmd.log): correct (7e-7 .. 1.4e-6 of the largest output vs the library's
// build) but no faster -- with one whole tile per worker (200 workers, no split) 2048 x 768 x 3200 takes 39.0 us against the
// library's 39.4: the 8-wave 256 x 128 pipeline runs a 1 536-MFMA-cycle chunk in ~2 600 cycles, like the 4-wave one; with
// 256 workers every tile is split and the 32 MB of partials written through and read back put 6 us ON TOP (45.3 us) instead of
// taking 22 % off.  A stream-K fix-up only pays once its partial volume (workers x tile bytes) is small against the launch:
// not at these sizes.  Kept as an experiment next to its harness (tools/gemm_bench.hip), outside the library.
// Workspace (d.splitk_ws): [FLAG_BYTES of flags, zero before the launch and left zero by it][workers x 128 KB of partials].
#pragma once
#include "../styletts2_amd/csrc/st2_conv_epilogue.h"

namespace st2sk {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef st2_f32x16 f32x16;
typedef st2_f32x4 f32x4;

constexpr int NT = 512;  // 8 waves
constexpr int WM = 8, TN = 4;
constexpr int BM = 32 * WM, BN = 32 * TN;  // 256 x 128 output tile
constexpr int FLAG_BYTES = 4096;
constexpr int MAX_WORKERS = FLAG_BYTES / 4;
constexpr int64_t PART_FLOATS = (int64_t)BM * BN;
constexpr int NSET = 3;  // weight-fragment register sets = prefetch distance + 1

__device__ __forceinline__ void store16_sc1(float* p, f32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}

template <int CI_T>
__global__ __launch_bounds__(NT, 1) void gemm_sk_kernel(const st2_conv_desc d, const int workers, const int mt, const int nt,
                                                         unsigned* flags, float* parts, int* status) {
  constexpr int CG = CI_T / 8;    // 8-channel groups per chunk
  constexpr int ROWS = 2 * CG;    // staged rows per chunk: (plane, group)
  constexpr int S16 = CI_T / 16;  // MFMA k-steps per chunk
  constexpr int XW = BN;          // k = 1: no halo columns
  constexpr int LBUF = ROWS * XW;
  constexpr int NS = LBUF / NT;  // 16-byte slots per thread per chunk
  static_assert(LBUF % NT == 0, "chunk image must be a whole number of slots per thread");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  h8* lds = reinterpret_cast<h8*>(smem_raw);  // [2 buffers][ROWS][XW] slots of 16 B

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kg = lane >> 5, l31 = lane & 31;
  const int nchunk = d.wq_cin_pad / CI_T;
  const int64_t tiles = (int64_t)mt * nt * d.B;
  const int64_t units = tiles * nchunk;
  const int w = blockIdx.x;
  auto first_unit = [&](int k) { return units * k / workers; };
  const int64_t u0 = first_unit(w), u1 = first_unit(w + 1);
  if (u0 >= u1) return;

  const int Lp = d.xs_lp;
  const int64_t gplane = (int64_t)d.xs_cg * Lp;
  int soff[NS];
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const int slot = tid + i * NT;
    const int row = slot / XW, col = slot - row * XW;
    soff[i] = (int)((row / CG) * gplane + (int64_t)(row % CG) * Lp + col);
  }
  const int64_t a_step = (int64_t)2 * d.wq_co_pad * 2;  // h8 units per k-step
  float* my_part = parts + (int64_t)w * PART_FLOATS + ((int64_t)wave * 16 * 64 + lane) * 4;

  auto run_segment = [&](const int64_t t, const int c0, const int c1, const bool finish) __attribute__((always_inline)) {
    const int mI = (int)(t % mt);
    const int64_t r = t / mt;
    const int nI = (int)(r % nt), b = (int)(r / nt);
    const int m0 = mI * BM, n0 = nI * BN;
    f32x16 acc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
    const h8* xsb = reinterpret_cast<const h8*>(d.xs) + (int64_t)b * 2 * gplane + (n0 + d.xs_halo);
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
    const int co_a = min(m0 + wave * 32 + l31, d.wq_co_pad - 1);  // rows past the packing re-read its last row (never stored)
    const h8* ap = reinterpret_cast<const h8*>(d.wq) + ((int64_t)kg * d.wq_co_pad + co_a) * 2 + (int64_t)c0 * S16 * a_step;
    const int nsteps = (c1 - c0) * S16;
    h8 a_hi[NSET], a_lo[NSET];
    load_chunk(c0);
#pragma unroll
    for (int k = 0; k < NSET - 1; ++k) {
      if (k > 0 && k < nsteps) ap += a_step;
      a_hi[k] = ap[0];
      a_lo[k] = ap[1];
    }
    store_chunk(0);
    __syncthreads();
    int done = NSET - 1;  // k-steps whose weights have been requested
    __builtin_amdgcn_s_setprio(1);
    for (int c = c0; c < c1; ++c) {
      const int buf = (c - c0) & 1;
      const bool more = c + 1 < c1;
      const h8* xbuf = lds + (size_t)buf * LBUF + kg * XW + l31;
#pragma unroll
      for (int s = 0; s < S16; ++s) {
        const int cur = s % NSET, pre = (s + NSET - 1) % NSET;
        if (done < nsteps) ap += a_step;  // scalar select, no branch around the loads (the last fragment is re-read)
        ++done;
        a_hi[pre] = ap[0];
        a_lo[pre] = ap[1];
        if (s == 0) load_chunk(more ? c + 1 : c);
        if (s == S16 - 1) store_chunk(buf ^ 1);
        __builtin_amdgcn_sched_barrier(0x786);  // neither VMEM nor MFMA crosses: the prefetch distance is kept
        const h8 ah = a_hi[cur], al = a_lo[cur];
        const h8* xp = xbuf + (2 * s) * XW;
        h8 bh[TN], bl[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          bh[j] = xp[j * 32];
          bl[j] = xp[CG * XW + j * 32];
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], ah, acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[j], ah, acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], al, acc[j], 0, 0, 0);
      }
      if constexpr (S16 % NSET != 0) {  // the next chunk indexes its steps from 0 again: rotate the live sets
        h8 th[NSET], tl[NSET];
#pragma unroll
        for (int k = 0; k < NSET; ++k) {
          th[k] = a_hi[k];
          tl[k] = a_lo[k];
        }
#pragma unroll
        for (int k = 0; k < NSET; ++k) {
          a_hi[k] = th[(k + S16) % NSET];
          a_lo[k] = tl[(k + S16) % NSET];
        }
      }
      __syncthreads();
    }
    __builtin_amdgcn_s_setprio(0);

    if (!finish) {
      // publish the raw accumulators: [wave][16 quads][lane] x 16 bytes, write-through, then the flag
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 v = {acc[j][4 * q], acc[j][4 * q + 1], acc[j][4 * q + 2], acc[j][4 * q + 3]};
          store16_sc1(my_part + (j * 4 + q) * 64 * 4, v);
        }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_store(flags + w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    if (c0 > 0) {  // the head of this tile lives with lower-numbered workers: add their partials, nearest first
      for (int p = w - 1; p >= 0; --p) {
        if (tid == 0) {
          int spins = 0;
          while (__hip_atomic_load(flags + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
            __builtin_amdgcn_s_sleep(16);
            if (++spins > (1 << 22)) {  // ~seconds: the producer never ran (not co-resident and never scheduled)
              st2_raise_status(status, ST2_STATUS_GEMM_TIMEOUT);
              break;
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        const float* pp = parts + (int64_t)p * PART_FLOATS + ((int64_t)wave * 16 * 64 + lane) * 4;
#pragma unroll
        for (int jh = 0; jh < TN; jh += 2) {
          f32x4 v[2][4];
#pragma unroll
          for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int q = 0; q < 4; ++q) v[jj][q] = *reinterpret_cast<const f32x4*>(pp + ((jh + jj) * 4 + q) * 64 * 4);
#pragma unroll
          for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
              for (int e = 0; e < 4; ++e) acc[jh + jj][4 * q + e] += v[jj][q][e];
        }
        __syncthreads();  // everyone has read partial p: its flag may go back to zero (the launch leaves the flags clean)
        if (tid == 0) __hip_atomic_store(flags + p, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (first_unit(p) <= t * nchunk) break;  // worker p held the tile's first chunk
      }
    }
    st2_conv_epilogue<TN, WM, 1>(d, acc, b, m0, n0, wave, 0, l31, kg);
  };

  // Segment order: the open tail of the range first (a partial someone else waits for), then the finishing segments.
  const int64_t t_first = u0 / nchunk, t_last = (u1 - 1) / nchunk;
  const bool tail_open = u1 < (t_last + 1) * nchunk;  // the range ends inside tile t_last: that segment is a partial
  const int n_fin = (int)(t_last - t_first) + (tail_open ? 0 : 1);
  for (int k = tail_open ? -1 : 0; k < n_fin; ++k) {  // ONE call site: one copy of the k loop and the epilogue
    const int64_t t = k < 0 ? t_last : t_first + k;
    const int64_t base = t * nchunk;
    run_segment(t, (int)((u0 > base ? u0 : base) - base), k < 0 ? (int)(u1 - base) : nchunk, k >= 0);
  }
}

// Worker count and workspace of a launch (0 workers = this geometry does not take the stream-K build).
inline int pick_workers(const st2_conv_desc& d, int num_cu, int ci_t) {
  if (d.ks != 1 || d.C_out < BM || d.wq_cin_pad % ci_t != 0 || d.part || d.pad_left != 0) return 0;
  const int64_t tiles = (int64_t)st2_cdiv(d.C_out, BM) * st2_cdiv(d.L_out, BN) * d.B;
  const int64_t units = tiles * (d.wq_cin_pad / ci_t);
  // worth it when the one-tile-per-workgroup builds leave CUs idle (less than ~3 rounds of 128 x 128 tiles) and every
  // worker still gets a few chunks
  const int64_t wg128 = (int64_t)st2_cdiv(d.C_out, 128) * st2_cdiv(d.L_out, 128) * d.B;
  if (wg128 >= 3 * (int64_t)num_cu || units < 4 * (int64_t)num_cu) return 0;
  return num_cu < MAX_WORKERS ? num_cu : MAX_WORKERS;
}
inline int64_t workspace_bytes(int workers) { return workers > 0 ? FLAG_BYTES + (int64_t)workers * PART_FLOATS * 4 : 0; }

template <int CI_T>
int launch(const st2_conv_desc& d, int workers, hipStream_t s) {
  ST2_REQUIRE(d.splitk_ws && d.splitk_ws_bytes >= workspace_bytes(workers), "st2_conv1d_xs (stream-K): workspace of %lld "
              "bytes needed, %lld given", (long long)workspace_bytes(workers), (long long)d.splitk_ws_bytes);
  ST2_REQUIRE((reinterpret_cast<uintptr_t>(d.splitk_ws) & 15) == 0, "st2_conv1d_xs (stream-K): workspace must be 16-byte aligned");
  ST2_REQUIRE(d.xs_cg * 8 >= d.wq_cin_pad, "st2_conv1d_xs: xs has %d channel groups, kernel needs %d", d.xs_cg, d.wq_cin_pad / 8);
  const int mt = st2_cdiv(d.C_out, BM), nt = st2_cdiv(d.L_out, BN);
  ST2_REQUIRE((int64_t)(nt - 1) * BN + d.xs_halo + BN <= d.xs_lp, "st2_conv1d_xs: xs rows of %d slots are too short for "
              "L_out=%d", d.xs_lp, d.L_out);
  constexpr size_t smem = (size_t)2 * (2 * CI_T / 8) * BN * 16;
  static std::atomic<uint64_t> attr_done{0};
  if (st2_first_use_on_device(attr_done))
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_sk_kernel<CI_T>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
  unsigned* flags = reinterpret_cast<unsigned*>(d.splitk_ws);
  float* parts = reinterpret_cast<float*>(reinterpret_cast<char*>(d.splitk_ws) + FLAG_BYTES);
  hipLaunchKernelGGL((gemm_sk_kernel<CI_T>), dim3(workers), dim3(NT), smem, s, d, workers, mt, nt, flags, parts,
                     st2_status_device_ptr());
  ST2_CHECK_LAUNCH("st2_conv1d_xs (stream-K)");
  return 0;
}

}  // namespace st2sk
