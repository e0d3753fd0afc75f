// Do an MFMA stream and a VALU stream on the SAME SIMD overlap, and does it depend on where the accumulators live?
//
// Round 3 measured on the fused conv that "on one SIMD the MFMA stream and the prologue's VALU stream nearly ADD whichever wave
// issues them" (profiles/archive/r03/r03_fused_conv_study.md) -- with accumulators in ARCHITECTURAL VGPRs, which is what hipcc picks for
// kernels bounded to >= 2 waves per SIMD.  An MFMA reads and writes its 16 accumulator registers on every pass; from AGPRs that
// traffic would not compete with another wave's VALU operand reads.  This probe times, per workgroup of 8 waves (two per SIMD):
//   waves 0-3: `mi` iterations of 4 independent v_mfma_f32_32x32x16_f16 chains, accumulators in VGPRs ("v") or AGPRs ("a")
//   waves 4-7: `vi` iterations of 8 independent v_fma_f32 chains (or v_pk_fma_f32, plain encoding)
// alone and together: together ~ max(alone) = the pipes overlap; together ~ sum = they serialise.
//   ./mfma_valu_overlap [mi=4000] [vi=16000]
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_valu_overlap.hip -o tools/bin/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                           \
  do {                                                                                  \
    hipError_t e_ = (x);                                                                \
    if (e_ != hipSuccess) {                                                             \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                          \
    }                                                                                   \
  } while (0)

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));

// ACC: 0 = VGPR accumulators, 1 = AGPR accumulators.  VAL: 0 = scalar v_fma_f32, 1 = v_pk_fma_f32 (plain encoding).
template <int ACC, int VAL>
__global__ __launch_bounds__(512) void overlap_kernel(float* out, int mi, int vi) {
  const int wave = threadIdx.x >> 6;
  if (wave < 4) {
    if (mi <= 0) return;
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    h8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      a[i] = (_Float16)(0.001f * (threadIdx.x + i));
      b[i] = (_Float16)(0.002f * (threadIdx.x ^ i));
    }
    for (int i = 0; i < mi; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if constexpr (ACC)
          asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[j]) : "v"(a), "v"(b));
        else
          asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));
      }
    }
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) t += acc[j][r];
    if (t == 12345.678f) out[threadIdx.x] = t;
  } else {
    if (vi <= 0) return;
    f2 x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = f2{0.001f * (threadIdx.x + i), 0.002f * (threadIdx.x - i)};
    const f2 m = f2{0.9995f, -0.9995f}, c = f2{0.001f, -0.001f};
    for (int i = 0; i < vi; ++i) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if constexpr (VAL) {
          asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(m), "v"(c));
        } else {
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[k].x) : "v"(m.x), "v"(c.x));
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[k].y) : "v"(m.y), "v"(c.y));
        }
      }
    }
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += x[k].x + x[k].y;
    if (t == 12345.678f) out[threadIdx.x] = t;
  }
}

template <int ACC, int VAL>
static float run(float* out, int mi, int vi) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((overlap_kernel<ACC, VAL>), dim3(256), dim3(512), 0, 0, out, mi, vi);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((overlap_kernel<ACC, VAL>), dim3(256), dim3(512), 0, 0, out, mi, vi);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / 5 * 1e3f;
}

int main(int argc, char** argv) {
  const int mi = argc > 1 ? atoi(argv[1]) : 4000, vi = argc > 2 ? atoi(argv[2]) : 16000;
  float* out;
  CK(hipMalloc(&out, 4096));
  printf("mfma_valu_overlap: %d x 4 MFMAs per MFMA wave, %d x 16 fma per VALU wave, one workgroup of 8 waves per CU\n", mi, vi);
#define ROW(ACC, VAL, name)                                                                                         \
  {                                                                                                                 \
    const float m = run<ACC, VAL>(out, mi, 0), v = run<ACC, VAL>(out, 0, vi), b = run<ACC, VAL>(out, mi, vi);       \
    printf("%-44s MFMA alone %7.1f us, VALU alone %7.1f us, together %7.1f us  (max %.1f, sum %.1f)\n", name, m, v, b, \
           m > v ? m : v, m + v);                                                                                   \
  }
  ROW(0, 0, "VGPR accumulators, scalar v_fma_f32:");
  ROW(1, 0, "AGPR accumulators, scalar v_fma_f32:");
  ROW(0, 1, "VGPR accumulators, v_pk_fma_f32:");
  ROW(1, 1, "AGPR accumulators, v_pk_fma_f32:");
  return 0;
}
