// Minimal, self-contained probe (no library code, builds in seconds) for what tools/lstm_load_repro.hip established in visit r06a:
//   * the product's BiLSTM kernels differ in 600 / 600 calls next to ANY 32-column-tile build of the xs conv on another queue, in
//     0 / 600 next to the 64- / 128-column builds, and in 0 / 600 when the two queues own disjoint CUs;
//   * every word a victim with the same access pattern LOADS is right (0 bad loads in ~10^9): it is not the memory path;
//   * what differs is per-lane VALU state of 16 (or 32 / 48) consecutive lanes of a wave, lanes 48-63 in the single-CU kernel.
// What distinguishes the 32-column build inside a wave: ONE accumulator tile (TN = 1), i.e. its three MFMAs per k-step are
// back-to-back DEPENDENT (SrcC = the previous instruction's vDst).  This probe tests that directly:
//   victims (stream A, 2 workgroups x 256 threads, nothing but registers unless said):
//     fma        4 independent fmaf chains per lane, checkpoints every 64 iterations
//     lds_fma    the recurrence's inner loop: acc = fmaf(w[k], h[k] (LDS broadcast), acc) with register-resident w
//     gld_fma    the same with w streamed from global memory (the single-CU kernel's sweep), h constant
//   aggressors (stream B, short kernels of 4-wave workgroups filling every CU):
//     none | mfma16_dep1 (v_mfma_f32_32x32x16_f16, ONE accumulator: every MFMA depends on the previous one) | mfma16_dep1_gap (the
//     same chain in groups of 3 with LDS reads between the groups, as in the conv) | mfma16_dep2 / mfma16_ind4 (2 / 4 independent
//     accumulators interleaved) | mfma32_dep1 (v_mfma_f32_32x32x2_f32 chain) | mfma16x16_dep1 (v_mfma_f32_16x16x32_f16 chain)
//     | mfma16_vgpr_dep1 (the chain with its accumulator in architectural VGPRs, as hipcc allocates the conv's) / _nop (16 wait
//     states between dependent MFMAs) | valu (plain VALU fma chains: co-residency without the matrix pipe)
// Every victim launch is compared bitwise with an idle run of the same launch; a differing launch prints the first differing
// checkpoint and the lanes that differ there.
//   ./simd_hazard_repro [victims=all] [aggressors=all] [calls=200]
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/simd_hazard_repro.hip -o tools/bin/simd_hazard_repro
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CK(x)                                                                                  \
  do {                                                                                         \
    hipError_t e_ = (x);                                                                       \
    if (e_ != hipSuccess) {                                                                    \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));        \
      exit(1);                                                                                 \
    }                                                                                          \
  } while (0)

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int CHK = 24;   // checkpoints per victim launch
constexpr int NWG = 2;    // victim workgroups

__device__ __forceinline__ float seedf(uint32_t i) {
  uint32_t h = i * 2654435761u + 12345u;
  h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
  return ((int)(h & 0xffff) - 32768) * (1.f / 32768.f);
}

// out[wg][chk][tid][4]
__global__ __launch_bounds__(256) void victim_fma(float* out) {
  const int tid = threadIdx.x;
  float x0 = seedf(tid), x1 = seedf(tid + 256), x2 = seedf(tid + 512), x3 = seedf(tid + 768);
  const float a = 0.99951172f, b = seedf(tid + 1024) * 0.01f;
  for (int c = 0; c < CHK; ++c) {
#pragma unroll 8
    for (int i = 0; i < 256; ++i) {
      x0 = fmaf(x0, a, b);
      x1 = fmaf(x1, -a, b);
      x2 = fmaf(x2, a, -b);
      x3 = fmaf(x3, -a, -b);
    }
    float* o = out + (((size_t)blockIdx.x * CHK + c) * 256 + tid) * 4;
    o[0] = x0; o[1] = x1; o[2] = x2; o[3] = x3;
  }
}

__global__ __launch_bounds__(256) void victim_lds_fma(float* out) {
  __shared__ float hs[2][256];
  const int tid = threadIdx.x;
  float w[4][32];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int k = 0; k < 32; ++k) w[g][k] = seedf(tid * 131 + g * 32 + k) * 0.0625f;
  hs[0][tid] = seedf(tid + 7777);
  __syncthreads();
  for (int c = 0; c < CHK; ++c) {
    const float* hp = hs[c & 1];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int k = 0; k < 256; ++k) {
      const float hk = hp[k];
      a0 = fmaf(w[0][k & 31], hk, a0);
      a1 = fmaf(w[1][k & 31], hk, a1);
      a2 = fmaf(w[2][k & 31], hk, a2);
      a3 = fmaf(w[3][k & 31], hk, a3);
    }
    float* o = out + (((size_t)blockIdx.x * CHK + c) * 256 + tid) * 4;
    o[0] = a0; o[1] = a1; o[2] = a2; o[3] = a3;
    hs[(c + 1) & 1][tid] = 0.25f * (a0 - a1) + 0.125f * (a2 - a3);
    __syncthreads();
  }
}

// Variants of victim_lds_fma that separate its ingredients.  VAR 1: scalar v_fmac_f32 instead of the SLP-packed v_pk_fma_f32 (an
// empty asm after every k keeps the four chains apart); 2: packed, but the LDS reads of a block of 32 k are COMPLETE (lgkmcnt(0)) before
// its FMAs start and no read is in flight while VALU runs; 3: both; 4: the baseline plus the plain sum of every h value read (out[3]
// carries it instead of the fourth chain: was the LDS data wrong, or the FMA?).
template <int VAR>
__global__ __launch_bounds__(256) void victim_lds_fma_v(float* out) {
  __shared__ __attribute__((aligned(16))) float hs[2][256];
  const int tid = threadIdx.x;
  float w[4][32];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int k = 0; k < 32; ++k) w[g][k] = seedf(tid * 131 + g * 32 + k) * 0.0625f;
  hs[0][tid] = seedf(tid + 7777);
  __syncthreads();
  for (int c = 0; c < CHK; ++c) {
    const float* hp = hs[c & 1];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, hsum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 256; kb += 32) {
      float hk[32];
#pragma unroll
      for (int k = 0; k < 32; ++k) hk[k] = hp[kb + k];
      if constexpr (VAR == 2 || VAR == 3) {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        a0 = fmaf(w[0][k], hk[k], a0);
        a1 = fmaf(w[1][k], hk[k], a1);
        a2 = fmaf(w[2][k], hk[k], a2);
        if constexpr (VAR == 4) hsum += hk[k]; else a3 = fmaf(w[3][k], hk[k], a3);
        if constexpr (VAR == 1 || VAR == 3) asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
      }
      if constexpr (VAR == 2 || VAR == 3) __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (VAR == 4) a3 = hsum;
    float* o = out + (((size_t)blockIdx.x * CHK + c) * 256 + tid) * 4;
    o[0] = a0; o[1] = a1; o[2] = a2; o[3] = a3;
    hs[(c + 1) & 1][tid] = 0.25f * (a0 - a1) + 0.125f * a2;
    __syncthreads();
  }
}

// Register-only victims, ONE instruction form each (inline asm pins it): which encodings of the packed-f32 VALU ops are affected?
// 8 independent accumulator pairs per lane, 64 x 8 instructions per checkpoint, nothing else in the loop.
typedef float f2 __attribute__((ext_vector_type(2)));
template <int FORM>
__global__ __launch_bounds__(256) void victim_asm(float* out) {
  const int tid = threadIdx.x;
  f2 d[8], a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    d[i] = f2{seedf(tid * 16 + i), seedf(tid * 16 + i + 8)};
    a[i] = f2{seedf(tid * 16 + i + 4096) * 0.001f, seedf(tid * 16 + i + 5000) * 0.001f};
    b[i] = f2{seedf(tid * 16 + i + 8192), seedf(tid * 16 + i + 9000)};
    if (FORM == 4 || FORM == 7 || FORM == 14) b[i] = f2{1.f + b[i].x * 0.0001f, 1.f + b[i].y * 0.0001f};
    if (FORM == 5 || FORM == 6 || FORM == 15 || FORM == 16) b[i] = b[i] * 0.001f;
  }
  float s1 = seedf(blockIdx.x + 77), s2 = seedf(blockIdx.x + 78);
  for (int c = 0; c < CHK; ++c) {
    for (int it = 0; it < 64; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if constexpr (FORM == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(d[i]) : "v"(a[i]), "v"(b[i]));
        else if constexpr (FORM == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(d[i]) : "v"(a[i]), "v"(b[i]));
        else if constexpr (FORM == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(d[i]) : "v"(a[i]), "v"(b[i]));
        else if constexpr (FORM == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(d[i]) : "v"(a[i]), "v"(b[i]));
        else if constexpr (FORM == 4) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(d[i]) : "v"(b[i]));
        else if constexpr (FORM == 5) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(d[i]) : "v"(b[i]));
        else if constexpr (FORM == 6) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d[i]) : "v"(b[i]));
        else if constexpr (FORM == 7) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(d[i]) : "v"(b[i]));
        else if constexpr (FORM == 8) {
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(d[i].x) : "v"(a[i].x), "v"(b[i].x));
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(d[i].y) : "v"(a[i].y), "v"(b[i].x));
        } else if constexpr (FORM == 9) {
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(d[i]) : "v"(a[i]), "s"(f2{s1, s2}));
        } else if constexpr (FORM == 10) {  // the broadcast done by hand: plain encoding, duplicated operand in a register pair
          f2 bb = f2{b[i].x, b[i].x};
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(d[i]) : "v"(a[i]), "v"(bb));
        } else if constexpr (FORM == 11) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 neg_lo:[0,1,0] neg_hi:[0,1,0]" : "+v"(d[i]) : "v"(a[i]), "v"(b[i]));
        else if constexpr (FORM == 12) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0]" : "+v"(d[i]) : "v"(a[i]), "v"(b[i]));
        else if constexpr (FORM == 13) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,1]" : "+v"(d[i]) : "v"(a[i]), "v"(b[i]));
        else if constexpr (FORM == 14) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[0,1]" : "+v"(d[i]) : "v"(b[i]));
        else if constexpr (FORM == 15) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1]" : "+v"(d[i]) : "v"(b[i]));
        else if constexpr (FORM == 16) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "+v"(d[i]) : "v"(b[i]));
        else if constexpr (FORM == 17) asm volatile("v_pk_mov_b32 %0, %0, %0 op_sel:[1,0] op_sel_hi:[0,0]" : "+v"(d[i]));
        else if constexpr (FORM == 18) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "+v"(d[i]) : "v"(a[i]), "v"(b[i]));
      }
    }
    float* o = out + (((size_t)blockIdx.x * CHK + c) * 256 + tid) * 4;
    o[0] = d[0].x + d[1].x + d[2].x + d[3].x; o[1] = d[0].y + d[1].y + d[2].y + d[3].y;
    o[2] = d[4].x + d[5].x + d[6].x + d[7].x; o[3] = d[4].y + d[5].y + d[6].y + d[7].y;
  }
}

__global__ __launch_bounds__(256) void victim_gld_fma(const float* __restrict__ W, float* out) {
  const int tid = threadIdx.x;
  for (int c = 0; c < CHK; ++c) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 8
    for (int k = 0; k < 256; ++k) {
      const float hk = 0.001f * (float)((k * 7 + c) & 63);
      const float* wk = W + (size_t)k * 1024 + tid;
      a0 = fmaf(wk[0], hk, a0);
      a1 = fmaf(wk[256], hk, a1);
      a2 = fmaf(wk[512], hk, a2);
      a3 = fmaf(wk[768], hk, a3);
    }
    asm volatile("" ::: "memory");
    float* o = out + (((size_t)blockIdx.x * CHK + c) * 256 + tid) * 4;
    o[0] = a0; o[1] = a1; o[2] = a2; o[3] = a3;
  }
}

// Self-checking LDS readers: LDS word i holds pat(i); every read is compared in registers and a wrong word is logged with what
// arrived.  MODE 0: all lanes read the SAME address (the recurrence's h broadcast), b32; 1: the same, b128; 2: lane-distinct
// consecutive addresses, b32; 3: lane-distinct, b128 (the conv's fragment reads).
struct LdsRec { uint32_t wg, tid, iter, idx, got, want, again, pad; };
__device__ __forceinline__ uint32_t pat(uint32_t i) { return i * 2654435761u + 0x9e3779b9u; }

template <int MODE>
__global__ __launch_bounds__(256) void victim_lds_chk(LdsRec* log, int* nlog, int maxlog, uint32_t* sink) {
  __shared__ __attribute__((aligned(16))) uint32_t hs[2048];
  const int tid = threadIdx.x;
  for (int i = tid; i < 2048; i += 256) hs[i] = pat(i);
  __syncthreads();
  uint32_t acc = 0;
  typedef __attribute__((address_space(3))) uint32_t lds_u32;
  const uint32_t base = (uint32_t)(uintptr_t)(lds_u32*)hs;  // byte offset of hs in the workgroup's LDS
  for (int it = 0; it < 96; ++it) {
#pragma unroll 4
    for (int k = 0; k < 256; k += 4) {
      uint32_t v[4], idx[4];
      if constexpr (MODE == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          idx[q] = (k + q + it) & 2047;
          asm volatile("ds_read_b32 %0, %1" : "=v"(v[q]) : "v"(base + 4 * idx[q]));
        }
      } else if constexpr (MODE == 1) {
        const uint32_t b = ((k + 4 * it) & 2047) & ~3u;
        u32x4 t;
        asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"(base + 4 * b));
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t));
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
#pragma unroll
        for (int q = 0; q < 4; ++q) idx[q] = b + q;
      } else if constexpr (MODE == 2) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          idx[q] = (tid + 64 * q + k + it) & 2047;
          asm volatile("ds_read_b32 %0, %1" : "=v"(v[q]) : "v"(base + 4 * idx[q]));
        }
      } else {
        const uint32_t b = (4 * tid + k + 4 * it) & 2047 & ~3u;
        u32x4 t;
        asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"(base + 4 * b));
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t));
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
#pragma unroll
        for (int q = 0; q < 4; ++q) idx[q] = b + q;
      }
      if constexpr (MODE == 0 || MODE == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        acc += v[q];
        if (v[q] != pat(idx[q])) {
          const int slot = atomicAdd(nlog, 1);
          if (slot < maxlog) {
            LdsRec r;
            r.wg = blockIdx.x; r.tid = tid; r.iter = it * 256 + k + q; r.idx = idx[q]; r.got = v[q]; r.want = pat(idx[q]);
            r.again = hs[idx[q]];
            r.pad = 0;
            log[slot] = r;
          }
        }
      }
    }
  }
  if (acc == 0x1234567u) sink[0] = acc;
}

// ---- aggressors ---------------------------------------------------------------------------------------------------------------------
template <int NACC, int GAP>
__global__ __launch_bounds__(256) void aggr_mfma16(float* out, int iters) {
  __shared__ h8 pad[256];
  f32x16 acc[NACC];
#pragma unroll
  for (int j = 0; j < NACC; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  h8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = (_Float16)(0.001f * (threadIdx.x + i));
    b[i] = (_Float16)(0.002f * (threadIdx.x ^ i));
  }
  pad[threadIdx.x] = a;
  __syncthreads();
  for (int i = 0; i < iters; ++i) {
    if constexpr (GAP) {  // groups of three dependent MFMAs with two LDS reads in front, as in the 32-column conv build
      a = pad[(threadIdx.x + i) & 255];
      b = pad[(threadIdx.x + 2 * i + 1) & 255];
#pragma unroll
      for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, a, acc[j], 0, 0, 0);
    } else {
#pragma unroll
      for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
    }
  }
  float t = 0.f;
#pragma unroll
  for (int j = 0; j < NACC; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) t += acc[j][r];
  if (t == 12345.678f) out[threadIdx.x] = t;
}

// The same dependent chain with the accumulator in ARCHITECTURAL VGPRs (what hipcc gives the conv kernel: v[0:15]) instead of
// AGPRs (what it gives the simple loops above): inline asm pins the register class.  NOPS = s_nop wait states between two
// dependent MFMAs (0 = back to back, as in the conv's groups of three).
template <int NOPS>
__global__ __launch_bounds__(256) void aggr_mfma16_vgpr(float* out, int iters) {
  __shared__ h8 pad[256];
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  h8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = (_Float16)(0.001f * (threadIdx.x + i));
    b[i] = (_Float16)(0.002f * (threadIdx.x ^ i));
  }
  pad[threadIdx.x] = a;
  __syncthreads();
  for (int i = 0; i < iters; ++i) {
    a = pad[(threadIdx.x + i) & 255];
    b = pad[(threadIdx.x + 2 * i + 1) & 255];
    if constexpr (NOPS == 0) {
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_f16 %0, %2, %1, %0\n\t"
                   "v_mfma_f32_32x32x16_f16 %0, %1, %1, %0" : "+v"(acc) : "v"(a), "v"(b));
    } else {
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n\ts_nop 15\n\tv_mfma_f32_32x32x16_f16 %0, %2, %1, %0\n\ts_nop 15\n\t"
                   "v_mfma_f32_32x32x16_f16 %0, %1, %1, %0\n\ts_nop 15" : "+v"(acc) : "v"(a), "v"(b));
    }
  }
  float t = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) t += acc[r];
  if (t == 12345.678f) out[threadIdx.x] = t;
}

__global__ __launch_bounds__(256) void aggr_mfma32(float* out, int iters) {
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const float a = 0.001f * threadIdx.x, b = 0.002f * (threadIdx.x ^ 5);
  for (int i = 0; i < iters; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  float t = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) t += acc[r];
  if (t == 12345.678f) out[threadIdx.x] = t;
}

__global__ __launch_bounds__(256) void aggr_mfma16x16(float* out, int iters) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  h8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = (_Float16)(0.001f * (threadIdx.x + i));
    b[i] = (_Float16)(0.002f * (threadIdx.x ^ i));
  }
  for (int i = 0; i < iters; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
  const float t = acc[0] + acc[1] + acc[2] + acc[3];
  if (t == 12345.678f) out[threadIdx.x] = t;
}

// v_mfma_f32_16x16x32_f16 chains: ACCV = accumulator in architectural VGPRs (else whatever hipcc picks: AGPRs), NACC chains
// interleaved, NOPS wait states after every MFMA.
template <int ACCV, int NACC, int NOPS>
__global__ __launch_bounds__(256) void aggr_mfma16x16_v(float* out, int iters) {
  f32x4 acc[NACC];
#pragma unroll
  for (int j = 0; j < NACC; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  h8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = (_Float16)(0.001f * (threadIdx.x + i));
    b[i] = (_Float16)(0.002f * (threadIdx.x ^ i));
  }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < NACC; ++j) {
      if constexpr (ACCV) {
        if constexpr (NOPS)
          asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\ts_nop 7" : "+v"(acc[j]) : "v"(a), "v"(b));
        else
          asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));
      } else {
        if constexpr (NOPS)
          asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\ts_nop 7" : "+a"(acc[j]) : "v"(a), "v"(b));
        else
          asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[j]) : "v"(a), "v"(b));
      }
    }
  }
  float t = 0.f;
#pragma unroll
  for (int j = 0; j < NACC; ++j) t += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  if (t == 12345.678f) out[threadIdx.x] = t;
}

__global__ __launch_bounds__(256) void aggr_valu(float* out, int iters) {
  float x0 = seedf(threadIdx.x), x1 = x0 + 1.f, x2 = x0 - 1.f, x3 = x0 * 0.5f;
  for (int i = 0; i < iters; ++i) {
    x0 = fmaf(x0, 0.9995f, 0.001f);
    x1 = fmaf(x1, -0.9995f, 0.001f);
    x2 = fmaf(x2, 0.9995f, -0.001f);
    x3 = fmaf(x3, -0.9995f, -0.001f);
  }
  if (x0 + x1 + x2 + x3 == 12345.678f) out[threadIdx.x] = x0;
}

static std::vector<std::string> split(const char* s) {
  std::vector<std::string> out;
  std::string cur;
  for (const char* p = s; *p; ++p) {
    if (*p == ',') { out.push_back(cur); cur.clear(); } else cur.push_back(*p);
  }
  if (!cur.empty()) out.push_back(cur);
  return out;
}

int main(int argc, char** argv) {
  const std::vector<std::string> all_v = {"asm_pk_fma", "asm_pk_fma_sel1lo", "asm_pk_fma_sel1hi", "asm_pk_fma_sel0lo", "asm_pk_mul_sel", "asm_pk_add_sel",
                                          "asm_pk_add", "asm_pk_mul", "asm_fma", "asm_pk_fma_sgpr_sel", "asm_pk_fma_dup", "asm_pk_fma_neg", "asm_pk_fma_sel0hi",
                                          "asm_pk_fma_sel2hi", "asm_pk_mul_selhi", "asm_pk_add_selhi", "asm_pk_add_swap", "asm_pk_mov_swap", "asm_pk_fma_sel1swap", "fma", "lds_fma", "lds_fma_scalar", "lds_fma_wait", "lds_fma_scalar_wait", "lds_fma_hsum", "gld_fma", "lds_bcast_b32", "lds_bcast_b128", "lds_lane_b32", "lds_lane_b128"};
  const std::vector<std::string> all_a = {"none", "mfma16_vgpr_dep1", "mfma16_vgpr_dep1_nop", "mfma16_dep1", "mfma16_dep1_gap", "mfma16_dep2", "mfma16_ind4", "mfma32_dep1",
                                          "mfma16x16_dep1", "mfma16x16_a_dep1", "mfma16x16_a_dep1_nop", "mfma16x16_a_ind2", "mfma16x16_a_ind4",
                                          "mfma16x16_v_dep1", "mfma16x16_v_dep1_nop", "mfma16x16_v_ind2", "valu"};
  std::vector<std::string> victims = (argc > 1 && strcmp(argv[1], "all")) ? split(argv[1]) : all_v;
  std::vector<std::string> aggrs = (argc > 2 && strcmp(argv[2], "all")) ? split(argv[2]) : all_a;
  const int calls = argc > 3 ? atoi(argv[3]) : 200;
  const int a_iters = argc > 4 ? atoi(argv[4]) : 400;
  const int a_grid = argc > 5 ? atoi(argv[5]) : 720;
  hipStream_t sa, sb;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  const size_t out_n = (size_t)NWG * CHK * 256 * 4;
  float *out, *W, *aout;
  CK(hipMalloc(&out, out_n * 4));
  CK(hipMalloc(&W, (size_t)256 * 1024 * 4));
  CK(hipMalloc(&aout, 4096));
  {
    std::vector<float> hw((size_t)256 * 1024);
    uint32_t st = 99u;
    for (auto& v : hw) { st = st * 1664525u + 1013904223u; v = ((int)(st >> 8 & 0xffff) - 32768) / 32768.f / 8.f; }
    CK(hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  }
  constexpr int MAXLOG = 2048;
  LdsRec* lrec;
  int* nlog;
  uint32_t* sink;
  CK(hipMalloc(&lrec, sizeof(LdsRec) * MAXLOG));
  CK(hipMalloc(&nlog, 4));
  CK(hipMalloc(&sink, 64));
  CK(hipMemset(nlog, 0, 4));
  auto launch_victim = [&](const std::string& v) {
    if (v == "fma") hipLaunchKernelGGL(victim_fma, dim3(NWG), dim3(256), 0, sa, out);
    else if (v == "lds_fma") hipLaunchKernelGGL(victim_lds_fma, dim3(NWG), dim3(256), 0, sa, out);
    else if (v.rfind("asm_", 0) == 0) {
      static const char* names[] = {"asm_pk_fma", "asm_pk_fma_sel1lo", "asm_pk_fma_sel1hi", "asm_pk_fma_sel0lo", "asm_pk_mul_sel", "asm_pk_add_sel",
                                    "asm_pk_add", "asm_pk_mul", "asm_fma", "asm_pk_fma_sgpr_sel", "asm_pk_fma_dup", "asm_pk_fma_neg",
                                    "asm_pk_fma_sel0hi", "asm_pk_fma_sel2hi", "asm_pk_mul_selhi", "asm_pk_add_selhi", "asm_pk_add_swap", "asm_pk_mov_swap",
                                    "asm_pk_fma_sel1swap"};
      int f = -1;
      for (int i = 0; i < 19; ++i) if (v == names[i]) f = i;
      const dim3 g(NWG), b(256);
      switch (f) {
        case 0: hipLaunchKernelGGL((victim_asm<0>), g, b, 0, sa, out); break;
        case 1: hipLaunchKernelGGL((victim_asm<1>), g, b, 0, sa, out); break;
        case 2: hipLaunchKernelGGL((victim_asm<2>), g, b, 0, sa, out); break;
        case 3: hipLaunchKernelGGL((victim_asm<3>), g, b, 0, sa, out); break;
        case 4: hipLaunchKernelGGL((victim_asm<4>), g, b, 0, sa, out); break;
        case 5: hipLaunchKernelGGL((victim_asm<5>), g, b, 0, sa, out); break;
        case 6: hipLaunchKernelGGL((victim_asm<6>), g, b, 0, sa, out); break;
        case 7: hipLaunchKernelGGL((victim_asm<7>), g, b, 0, sa, out); break;
        case 8: hipLaunchKernelGGL((victim_asm<8>), g, b, 0, sa, out); break;
        case 9: hipLaunchKernelGGL((victim_asm<9>), g, b, 0, sa, out); break;
        case 10: hipLaunchKernelGGL((victim_asm<10>), g, b, 0, sa, out); break;
        case 11: hipLaunchKernelGGL((victim_asm<11>), g, b, 0, sa, out); break;
        case 12: hipLaunchKernelGGL((victim_asm<12>), g, b, 0, sa, out); break;
        case 13: hipLaunchKernelGGL((victim_asm<13>), g, b, 0, sa, out); break;
        case 14: hipLaunchKernelGGL((victim_asm<14>), g, b, 0, sa, out); break;
        case 15: hipLaunchKernelGGL((victim_asm<15>), g, b, 0, sa, out); break;
        case 16: hipLaunchKernelGGL((victim_asm<16>), g, b, 0, sa, out); break;
        case 17: hipLaunchKernelGGL((victim_asm<17>), g, b, 0, sa, out); break;
        case 18: hipLaunchKernelGGL((victim_asm<18>), g, b, 0, sa, out); break;
        default: fprintf(stderr, "unknown victim %s\n", v.c_str()); exit(1);
      }
    }
    else if (v == "lds_fma_scalar") hipLaunchKernelGGL((victim_lds_fma_v<1>), dim3(NWG), dim3(256), 0, sa, out);
    else if (v == "lds_fma_wait") hipLaunchKernelGGL((victim_lds_fma_v<2>), dim3(NWG), dim3(256), 0, sa, out);
    else if (v == "lds_fma_scalar_wait") hipLaunchKernelGGL((victim_lds_fma_v<3>), dim3(NWG), dim3(256), 0, sa, out);
    else if (v == "lds_fma_hsum") hipLaunchKernelGGL((victim_lds_fma_v<4>), dim3(NWG), dim3(256), 0, sa, out);
    else if (v == "gld_fma") hipLaunchKernelGGL(victim_gld_fma, dim3(NWG), dim3(256), 0, sa, W, out);
    else if (v == "lds_bcast_b32") hipLaunchKernelGGL((victim_lds_chk<0>), dim3(NWG), dim3(256), 0, sa, lrec, nlog, MAXLOG, sink);
    else if (v == "lds_bcast_b128") hipLaunchKernelGGL((victim_lds_chk<1>), dim3(NWG), dim3(256), 0, sa, lrec, nlog, MAXLOG, sink);
    else if (v == "lds_lane_b32") hipLaunchKernelGGL((victim_lds_chk<2>), dim3(NWG), dim3(256), 0, sa, lrec, nlog, MAXLOG, sink);
    else if (v == "lds_lane_b128") hipLaunchKernelGGL((victim_lds_chk<3>), dim3(NWG), dim3(256), 0, sa, lrec, nlog, MAXLOG, sink);
    else { fprintf(stderr, "unknown victim %s\n", v.c_str()); exit(1); }
  };
  auto launch_aggr = [&](const std::string& a) {
    const dim3 g(a_grid), b(256);
    if (a == "mfma16_vgpr_dep1") hipLaunchKernelGGL((aggr_mfma16_vgpr<0>), g, b, 0, sb, aout, a_iters);
    else if (a == "mfma16_vgpr_dep1_nop") hipLaunchKernelGGL((aggr_mfma16_vgpr<15>), g, b, 0, sb, aout, a_iters);
    else if (a == "mfma16_dep1") hipLaunchKernelGGL((aggr_mfma16<1, 0>), g, b, 0, sb, aout, 3 * a_iters);
    else if (a == "mfma16_dep1_gap") hipLaunchKernelGGL((aggr_mfma16<1, 1>), g, b, 0, sb, aout, a_iters);
    else if (a == "mfma16_dep2") hipLaunchKernelGGL((aggr_mfma16<2, 0>), g, b, 0, sb, aout, 3 * a_iters / 2);
    else if (a == "mfma16_ind4") hipLaunchKernelGGL((aggr_mfma16<4, 0>), g, b, 0, sb, aout, 3 * a_iters / 4);
    else if (a == "mfma32_dep1") hipLaunchKernelGGL(aggr_mfma32, g, b, 0, sb, aout, 3 * a_iters);
    else if (a == "mfma16x16_dep1") hipLaunchKernelGGL(aggr_mfma16x16, g, b, 0, sb, aout, 6 * a_iters);
    else if (a == "mfma16x16_a_dep1") hipLaunchKernelGGL((aggr_mfma16x16_v<0, 1, 0>), g, b, 0, sb, aout, 6 * a_iters);
    else if (a == "mfma16x16_a_dep1_nop") hipLaunchKernelGGL((aggr_mfma16x16_v<0, 1, 1>), g, b, 0, sb, aout, 3 * a_iters);
    else if (a == "mfma16x16_a_ind2") hipLaunchKernelGGL((aggr_mfma16x16_v<0, 2, 0>), g, b, 0, sb, aout, 3 * a_iters);
    else if (a == "mfma16x16_a_ind4") hipLaunchKernelGGL((aggr_mfma16x16_v<0, 4, 0>), g, b, 0, sb, aout, 3 * a_iters / 2);
    else if (a == "mfma16x16_v_dep1") hipLaunchKernelGGL((aggr_mfma16x16_v<1, 1, 0>), g, b, 0, sb, aout, 6 * a_iters);
    else if (a == "mfma16x16_v_dep1_nop") hipLaunchKernelGGL((aggr_mfma16x16_v<1, 1, 1>), g, b, 0, sb, aout, 3 * a_iters);
    else if (a == "mfma16x16_v_ind2") hipLaunchKernelGGL((aggr_mfma16x16_v<1, 2, 0>), g, b, 0, sb, aout, 3 * a_iters);
    else if (a == "valu") hipLaunchKernelGGL(aggr_valu, g, b, 0, sb, aout, 20 * a_iters);
    else { fprintf(stderr, "unknown aggressor %s\n", a.c_str()); exit(1); }
  };
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  std::vector<float> ref(out_n), cur(out_n);
  printf("simd_hazard_repro: calls=%d aggressor iters=%d grid=%d\n", calls, a_iters, a_grid);
  for (auto& v : victims) {
    launch_victim(v);
    CK(hipStreamSynchronize(sa));
    CK(hipEventRecord(e0, sa));
    for (int i = 0; i < 10; ++i) launch_victim(v);
    CK(hipEventRecord(e1, sa));
    CK(hipEventSynchronize(e1));
    float tv = 0;
    CK(hipEventElapsedTime(&tv, e0, e1));
    tv /= 10;
    CK(hipMemcpy(ref.data(), out, out_n * 4, hipMemcpyDeviceToHost));
    for (auto& a : aggrs) {
      float ta = 0.02f;
      if (a != "none") {
        launch_aggr(a);
        CK(hipStreamSynchronize(sb));
        CK(hipEventRecord(e0, sb));
        for (int i = 0; i < 10; ++i) launch_aggr(a);
        CK(hipEventRecord(e1, sb));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ta, e0, e1));
        ta /= 10;
      }
      int bad = 0;
      std::string detail;
      const bool selfcheck = v.rfind("lds_bcast", 0) == 0 || v.rfind("lds_lane", 0) == 0;
      for (int i = 0; i < calls; ++i) {
        if (selfcheck) {
          const int n_aggr = a == "none" ? 0 : (int)(2.0f * tv / ta) + 4;
          for (int k = 0; k < n_aggr; ++k) launch_aggr(a);
          CK(hipMemsetAsync(nlog, 0, 4, sa));
          launch_victim(v);
          CK(hipStreamSynchronize(sa));
          int n = 0;
          CK(hipMemcpy(&n, nlog, 4, hipMemcpyDeviceToHost));
          if (n) {
            ++bad;
            if (detail.size() < 1500) {
              std::vector<LdsRec> r(std::min(n, MAXLOG));
              CK(hipMemcpy(r.data(), lrec, sizeof(LdsRec) * r.size(), hipMemcpyDeviceToHost));
              std::sort(r.begin(), r.end(), [](const LdsRec& x, const LdsRec& y) {
                return x.wg != y.wg ? x.wg < y.wg : (x.iter != y.iter ? x.iter < y.iter : x.tid < y.tid); });
              char buf[400];
              uint32_t inv = 1;
              for (int t2 = 0; t2 < 5; ++t2) inv *= 2u - 2654435761u * inv;
              int shown = 0;
              for (size_t q = 0; q < r.size() && shown < 4;) {
                size_t e = q + 1;
                while (e < r.size() && r[e].wg == r[q].wg && r[e].iter == r[q].iter && r[e].tid == r[e - 1].tid + 1) ++e;
                const uint32_t src = (r[q].got - 0x9e3779b9u) * inv;
                snprintf(buf, sizeof(buf), "    call %d: %d bad words; wg %u read %u lanes %u-%u idx %u: got %08x want %08x%s reread %s\n", i, n,
                         r[q].wg, r[q].iter, r[q].tid, r[e - 1].tid, r[q].idx, r[q].got, r[q].want,
                         src < 2048 ? (std::string(" (= LDS word ") + std::to_string(src) + ")").c_str() : "",
                         r[q].again == r[q].want ? "ok" : "BAD");
                detail += buf;
                ++shown;
                q = e;
              }
            }
          }
          continue;
        }
        const int n_aggr = a == "none" ? 0 : (int)(2.0f * tv / ta) + 4;
        for (int k = 0; k < n_aggr; ++k) launch_aggr(a);
        launch_victim(v);
        CK(hipStreamSynchronize(sa));
        CK(hipMemcpy(cur.data(), out, out_n * 4, hipMemcpyDeviceToHost));
        if (memcmp(cur.data(), ref.data(), out_n * 4)) {
          ++bad;
          if (detail.size() < 900) {
            for (int wg = 0; wg < NWG; ++wg) {
              int first_c = -1;
              for (int c = 0; c < CHK && first_c < 0; ++c)
                if (memcmp(&cur[((size_t)wg * CHK + c) * 1024], &ref[((size_t)wg * CHK + c) * 1024], 4096)) first_c = c;
              if (first_c < 0) continue;
              char buf[512];
              int comp = 0;
              for (int t = 0; t < 256; ++t)
                for (int q = 0; q < 4; ++q)
                  if (memcmp(&cur[(((size_t)wg * CHK + first_c) * 256 + t) * 4 + q], &ref[(((size_t)wg * CHK + first_c) * 256 + t) * 4 + q], 4))
                    comp |= 1 << q;
              int n = snprintf(buf, sizeof(buf), "    call %d wg %d: first differing checkpoint %d, outputs 0x%x, lanes", i, wg, first_c, comp);
              int lo = -1, prev = -2;
              for (int t = 0; t <= 256; ++t) {
                const bool d = t < 256 && memcmp(&cur[(((size_t)wg * CHK + first_c) * 256 + t) * 4],
                                                 &ref[(((size_t)wg * CHK + first_c) * 256 + t) * 4], 16);
                if (d && lo < 0) lo = t;
                if (!d && lo >= 0) {
                  if (n < 480) n += snprintf(buf + n, sizeof(buf) - n, " %d-%d", lo, prev);
                  lo = -1;
                }
                if (d) prev = t;
              }
              detail += buf;
              detail += "\n";
            }
          }
        }
      }
      CK(hipDeviceSynchronize());
      printf("victim=%-8s aggr=%-16s calls=%4d bad_calls=%4d  (victim %.1f us idle, aggressor %.1f us / launch idle)\n", v.c_str(),
             a.c_str(), calls, bad, tv * 1e3f, ta * 1e3f);
      if (!detail.empty()) fputs(detail.c_str(), stdout);
      fflush(stdout);
    }
  }
  return 0;
}
