// Normalisation statistics and the batched style FC.  All HBM-bound row/column reductions:
// coalesced reads, fp64 partial sums, fixed shuffle-tree order (bitwise reproducible).
#include "st2_common.h"

namespace {

// One workgroup per (b, c) row.  Rows are up to 48 001 samples (generator) or 400/800 (decoder front).
template <int NT>
__global__ __launch_bounds__(NT) void instnorm_stats_kernel(const float* __restrict__ x, int64_t x_bs, int x_cs,
                                                            int C, int L, float eps, float* __restrict__ stats) {
  const int row = blockIdx.x;
  const int b = row / C;
  const int c = row % C;
  const float* xr = x + (int64_t)b * x_bs + (int64_t)c * x_cs;
  double s = 0.0, ss = 0.0;
  if ((reinterpret_cast<uintptr_t>(xr) & 15) == 0) {
    // 16-byte loads, four independent fp64 chains per thread (round 6: one 4-byte load feeding one dependent fp64 add per iteration
    // ran the 48 001-sample rows at 2.6 TB/s); fixed order: component chains, then ((0 + 1) + (2 + 3)), then the tail
    typedef float f4 __attribute__((ext_vector_type(4)));
    double a[4] = {0.0, 0.0, 0.0, 0.0}, q[4] = {0.0, 0.0, 0.0, 0.0};
    const int L4 = L >> 2;
    const f4* x4 = reinterpret_cast<const f4*>(xr);
    for (int i = threadIdx.x; i < L4; i += NT) {
      const f4 v = x4[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const double d = (double)v[e];
        a[e] += d;
        q[e] += d * d;
      }
    }
    s = (a[0] + a[1]) + (a[2] + a[3]);
    ss = (q[0] + q[1]) + (q[2] + q[3]);
    for (int l = 4 * L4 + threadIdx.x; l < L; l += NT) {
      const double v = (double)xr[l];
      s += v;
      ss += v * v;
    }
  } else {
    for (int l = threadIdx.x; l < L; l += NT) {
      const double v = (double)xr[l];
      s += v;
      ss += v * v;
    }
  }
  s = st2_wave_sum(s);
  ss = st2_wave_sum(ss);
  __shared__ double red[2][NT / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    red[0][wave] = s;
    red[1][wave] = ss;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double ts = 0.0, tss = 0.0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) {
      ts += red[0][w];
      tss += red[1][w];
    }
    const double mean = ts / (double)L;
    double var = tss / (double)L - mean * mean;  // biased, InstanceNorm1d
    if (var < 0.0) var = 0.0;
    stats[(int64_t)row * 2 + 0] = (float)mean;
    stats[(int64_t)row * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

// LayerNorm over channels of an NCL tensor.  Workgroup = 64 positions x CW channel slices: lanes run along l
// (coalesced 256-byte rows), wave w sums channels w, w+CW, ... in fp64 with 8 independent loads in flight, and
// the CW partials are combined through LDS in a fixed order (bitwise reproducible).
constexpr int CW = 16;
__global__ __launch_bounds__(64 * CW) void colnorm_stats_kernel(const float* __restrict__ x, int64_t x_bs, int x_cs,
                                                                int C, int L, float eps, float* __restrict__ stats) {
  __shared__ double red[2][CW][64];
  const int lane = threadIdx.x & 63;
  const int w = threadIdx.x >> 6;
  const int l = blockIdx.x * 64 + lane;
  const int b = blockIdx.y;
  const bool live = l < L;
  const float* xb = x + (int64_t)b * x_bs + (live ? l : 0);
  double s = 0.0, ss = 0.0;
  int c = w;
  for (; c + 7 * CW < C; c += 8 * CW) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = xb[(int64_t)(c + u * CW) * x_cs];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const double dv = (double)v[u];
      s += dv;
      ss += dv * dv;
    }
  }
  for (; c < C; c += CW) {
    const double dv = (double)xb[(int64_t)c * x_cs];
    s += dv;
    ss += dv * dv;
  }
  red[0][w][lane] = s;
  red[1][w][lane] = ss;
  __syncthreads();
  if (w == 0 && live) {
    double ts = 0.0, tss = 0.0;
#pragma unroll
    for (int i = 0; i < CW; ++i) {
      ts += red[0][i][lane];
      tss += red[1][i][lane];
    }
    const double mean = ts / (double)C;
    double var = tss / (double)C - mean * mean;
    if (var < 0.0) var = 0.0;
    float* o = stats + ((int64_t)b * L + l) * 2;
    o[0] = (float)mean;
    o[1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

// h[b][j] = act(bias[j] + sum_k s[b][k] * wt[k*J + j]).  The layers behind this are tiny GEMMs (B <= 32 rows) with
// K = 128..1024: with one thread per output and a K-long serial loop the kernel was latency bound (16 workgroups,
// ~100 us for the denoiser's 1024x1024 mapping layers).  Workgroup = 64 outputs x 4 k-slices: thread (jj, ks) sweeps
// k in [ks*K/4, (ks+1)*K/4) for 8 batch rows (style vectors in LDS, read as 16-byte broadcasts; four independent
// coalesced weight loads in flight), the four partials are combined through LDS in slice order (fixed order =>
// bitwise reproducible).
// Latency, not bandwidth, bounds these mat-vecs (B <= 32 rows against a <= 2048 x J matrix, 41 launches per bench step): a
// thread walks its k slice with dependent-free but serially issued row loads, so the launch takes (k per thread / loads in
// flight) L2 round trips.  16 k slices per workgroup (1024 threads; round 2: 4 slices, 256 threads) cut a K = 1024 walk
// from 64 to 16 iterations of 4 row loads: 40 -> ~15 us per launch.  Partial sums meet in LDS in a fixed order.
constexpr int FC_BT = 8;
constexpr int FC_J = 64;
constexpr int FC_KS = 16;
constexpr int FC_NT = FC_J * FC_KS;
__global__ __launch_bounds__(FC_NT) void style_fc_kernel(const float* __restrict__ s, int B, int K,
                                                       const float* __restrict__ wt, const float* __restrict__ bias,
                                                       int J, int act, float* __restrict__ h) {
  extern __shared__ __attribute__((aligned(16))) float sl[];  // [FC_BT][K] then [FC_KS][FC_BT][FC_J]
  float* red = sl + FC_BT * K;
  const int jj = threadIdx.x & (FC_J - 1);
  const int ks = threadIdx.x / FC_J;
  const int j = blockIdx.x * FC_J + jj;
  const int jc = min(j, J - 1);
  const int b0 = blockIdx.y * FC_BT;
  const int nb = min(FC_BT, B - b0);
  for (int e = threadIdx.x; e < FC_BT * K; e += FC_NT) {
    const int bb = e / K, k = e % K;
    sl[e] = bb < nb ? s[(int64_t)(b0 + bb) * K + k] : 0.f;
  }
  __syncthreads();
  float acc[FC_BT];
#pragma unroll
  for (int bb = 0; bb < FC_BT; ++bb) acc[bb] = 0.f;
  const int kq = ((K + 3) / 4 + FC_KS - 1) / FC_KS * 4;  // k per slice, a multiple of 4, FC_KS * kq >= K
  const int k0 = ks * kq, k1 = min(K, k0 + kq);
  int k = k0;
  for (; k + 4 <= k1; k += 4) {
    const float w0 = wt[(int64_t)k * J + jc], w1 = wt[(int64_t)(k + 1) * J + jc];
    const float w2 = wt[(int64_t)(k + 2) * J + jc], w3 = wt[(int64_t)(k + 3) * J + jc];
#pragma unroll
    for (int bb = 0; bb < FC_BT; ++bb) {
      const float* sp = &sl[bb * K + k];
      float s0, s1, s2, s3;
      if ((K & 3) == 0) {
        const float4 sv = *reinterpret_cast<const float4*>(sp);
        s0 = sv.x; s1 = sv.y; s2 = sv.z; s3 = sv.w;
      } else {
        s0 = sp[0]; s1 = sp[1]; s2 = sp[2]; s3 = sp[3];
      }
      acc[bb] = fmaf(s0, w0, acc[bb]);
      acc[bb] = fmaf(s1, w1, acc[bb]);
      acc[bb] = fmaf(s2, w2, acc[bb]);
      acc[bb] = fmaf(s3, w3, acc[bb]);
    }
  }
  for (; k < k1; ++k) {
    const float w = wt[(int64_t)k * J + jc];
#pragma unroll
    for (int bb = 0; bb < FC_BT; ++bb) acc[bb] = fmaf(sl[bb * K + k], w, acc[bb]);
  }
#pragma unroll
  for (int bb = 0; bb < FC_BT; ++bb) red[(ks * FC_BT + bb) * FC_J + jj] = acc[bb];
  __syncthreads();
  // thread (jj, ks < FC_BT) finishes batch row ks: the k slices are summed in slice order
  if (j < J && ks < FC_BT) {
    const float bj = bias ? bias[j] : 0.f;
    const int bb = ks;
    if (bb < nb) {
      float v = red[(0 * FC_BT + bb) * FC_J + jj];
#pragma unroll
      for (int q = 1; q < FC_KS; ++q) v += red[(q * FC_BT + bb) * FC_J + jj];
      v += bj;
      if (act == ST2_ACT_GELU) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
      h[(int64_t)(b0 + bb) * J + j] = v;
    }
  }
}

// y = act((x - mean[b,l]) * rstd[b,l] * G[b,c] + Bt[b,c]), zero for l >= len[b].  HBM-bound elementwise pass; lanes run
// along l (coalesced rows), one thread handles 4 channels so the per-position statistics are loaded once per 4 outputs.
__global__ __launch_bounds__(256) void colnorm_apply_kernel(const float* __restrict__ x, int64_t x_bs, int x_cs,
                                                            const float* __restrict__ stats,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, int64_t gb_bs, int plus_one,
                                                            int act, float slope, const int* __restrict__ len,
                                                            float* __restrict__ y, int64_t y_bs, int y_cs, int C, int L) {
  const int l = blockIdx.x * 256 + threadIdx.x;
  const int c0 = blockIdx.y * 4;
  const int b = blockIdx.z;
  if (l >= L) return;
  const bool live = !len || l < len[b];
  const float2 st = reinterpret_cast<const float2*>(stats)[(int64_t)b * L + l];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = c0 + e;
    if (c >= C) break;
    float v = 0.f;
    if (live) {
      const float g0 = gamma[(int64_t)b * gb_bs + c];
      const float g = plus_one ? 1.0f + g0 : g0;
      const float u = (x[(int64_t)b * x_bs + (int64_t)c * x_cs + l] - st.x) * st.y;
      v = u * g + beta[(int64_t)b * gb_bs + c];
      if (act == ST2_ACT_LEAKY) v = v >= 0.f ? v : v * slope;
    }
    y[(int64_t)b * y_bs + (int64_t)c * y_cs + l] = v;
  }
}

}  // namespace

extern "C" int st2_colnorm_apply(const float* x, int64_t x_bs, int32_t x_cs, const float* stats, const float* gamma,
                                 const float* beta, int64_t gb_bs, int32_t gamma_plus_one, int32_t act, float slope,
                                 const int32_t* len, float* y, int64_t y_bs, int32_t y_cs, int32_t B, int32_t C,
                                 int32_t L, void* stream) {
  ST2_REQUIRE(x && stats && gamma && beta && y && B > 0 && C > 0 && L > 0, "st2_colnorm_apply: bad arguments");
  ST2_REQUIRE(act == ST2_ACT_NONE || act == ST2_ACT_LEAKY, "st2_colnorm_apply: act must be NONE or LEAKY");
  ST2_REQUIRE(B <= 65535 && (C + 3) / 4 <= 65535, "st2_colnorm_apply: grid too large");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(colnorm_apply_kernel, dim3(st2_cdiv(L, 256), st2_cdiv(C, 4), B), dim3(256), 0, s, x, x_bs, x_cs,
                     stats, gamma, beta, gb_bs, gamma_plus_one, act, slope, reinterpret_cast<const int*>(len), y, y_bs,
                     y_cs, C, L);
  ST2_CHECK_LAUNCH("st2_colnorm_apply");
  return 0;
}

extern "C" int st2_instnorm_stats(const float* x, int64_t x_bs, int32_t x_cs, int32_t B, int32_t C, int32_t L,
                                  float eps, float* stats, void* stream) {
  ST2_REQUIRE(x && stats && B > 0 && C > 0 && L > 0, "st2_instnorm_stats: bad arguments");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int rows = B * C;
  if (L <= 2048)
    hipLaunchKernelGGL((instnorm_stats_kernel<64>), dim3(rows), dim3(64), 0, s, x, x_bs, x_cs, C, L, eps, stats);
  else
    hipLaunchKernelGGL((instnorm_stats_kernel<256>), dim3(rows), dim3(256), 0, s, x, x_bs, x_cs, C, L, eps, stats);
  ST2_CHECK_LAUNCH("st2_instnorm_stats");
  return 0;
}

extern "C" int st2_colnorm_stats(const float* x, int64_t x_bs, int32_t x_cs, int32_t B, int32_t C, int32_t L,
                                 float eps, float* stats, void* stream) {
  ST2_REQUIRE(x && stats && B > 0 && C > 0 && L > 0, "st2_colnorm_stats: bad arguments");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(colnorm_stats_kernel, dim3(st2_cdiv(L, 64), B), dim3(64 * CW), 0, s, x, x_bs, x_cs, C, L, eps,
                     stats);
  ST2_CHECK_LAUNCH("st2_colnorm_stats");
  return 0;
}

extern "C" int st2_style_fc(const float* sv, int32_t B, int32_t K, const float* wt, const float* bias, int32_t J,
                            int32_t act, float* h, void* stream) {
  ST2_REQUIRE(sv && wt && h && B > 0 && K > 0 && J > 0, "st2_style_fc: bad arguments");
  ST2_REQUIRE(K <= 2048, "st2_style_fc: K=%d too large", K);
  ST2_REQUIRE(act == ST2_ACT_NONE || act == ST2_ACT_GELU, "st2_style_fc: act must be NONE or GELU");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const size_t smem = ((size_t)FC_BT * K + (size_t)FC_KS * FC_BT * FC_J) * sizeof(float);
  static std::atomic<uint64_t> attr_done{0};  // one bit per device ordinal
  st2_once_per_device(attr_done, [&] {  // up to 8 x 2048 staged inputs + 16 x 8 x 64 partial sums: 96 KB
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&style_fc_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
  });
  hipLaunchKernelGGL(style_fc_kernel, dim3(st2_cdiv(J, FC_J), st2_cdiv(B, FC_BT)), dim3(FC_NT), smem, s, sv, B, K, wt,
                     bias, J, act, h);
  ST2_CHECK_LAUNCH("st2_style_fc");
  return 0;
}
