// Activation / split helpers shared by the fused conv prologue (st2_conv1d_f16s.hip), the stand-alone activation pass
// (st2_actsplit.hip) and the conv epilogues.  Device-only, gfx950.
#pragma once
#include <hip/hip_runtime.h>

typedef _Float16 st2_h8 __attribute__((ext_vector_type(8)));

static __device__ __forceinline__ float leaky(float v, float slope) { return v >= 0.f ? v : v * slope; }

// Largest finite f16 magnitude; operands of the split-f16 convs are clamped to it before the hi cast.  One v_med3_f32 (the
// compare / select form costs four VALU instructions and two VCC wait states per element of a prologue that is VALU-bound:
// DESIGN.md section 3 iv); finite values map exactly as before, a NaN operand becomes -65504 -- like an overflow it differs
// from the clamped value, so the caller's `clamped != value` test reports it through ST2_STATUS_F16_RANGE.
static __device__ __forceinline__ float st2_clamp_f16(float u) { return __builtin_amdgcn_fmed3f(u, -65504.f, 65504.f); }

// sin(x)^2 to ~2.4e-7 absolute: n = rint(x/pi), r = x - n*pi (two-term, fma-exact), odd minimax
// polynomial of degree 9 on [-pi/2, pi/2] (max abs error 1.2e-7, fitted in tools/fit_sin.py).
static __device__ __forceinline__ float sin_sq(float x) {
  const float n = rintf(x * 0.3183098861837907f);
  float r = fmaf(n, -3.1415927410125732f, x);
  r = fmaf(n, 8.742277657347586e-08f, r);
  const float r2 = r * r;
  float p = 2.6000539037340786e-06f;
  p = fmaf(p, r2, -0.00019806614727713168f);
  p = fmaf(p, r2, 0.008333017118275166f);
  p = fmaf(p, r2, -0.16666656732559204f);
  const float s = fmaf(r2 * r, p, r);
  return s * s;
}

// sin(x) to 1.2e-7 absolute (same reduction and polynomial, sign restored from the parity of n)
static __device__ __forceinline__ float sin_acc(float x) {
  const float n = rintf(x * 0.3183098861837907f);
  float r = fmaf(n, -3.1415927410125732f, x);
  r = fmaf(n, 8.742277657347586e-08f, r);
  const float r2 = r * r;
  float p = 2.6000539037340786e-06f;
  p = fmaf(p, r2, -0.00019806614727713168f);
  p = fmaf(p, r2, 0.008333017118275166f);
  p = fmaf(p, r2, -0.16666656732559204f);
  const float s = fmaf(r2 * r, p, r);
  return ((int)n & 1) ? -s : s;
}

static __device__ __forceinline__ float snake(float v, float alpha, float inv_alpha) {
  return v + inv_alpha * sin_sq(alpha * v);  // x + (1/a) * sin(a*x)^2, Modules/istftnet.py:69
}

static __device__ __forceinline__ float gelu_erf(float v) {
  return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
}


// HF "gelu_new" (tanh approximation), the ALBERT FFN activation
static __device__ __forceinline__ float gelu_tanh(float v) {
  return 0.5f * v * (1.0f + tanhf(0.7978845608028654f * (v + 0.044715f * (v * v * v))));
}
