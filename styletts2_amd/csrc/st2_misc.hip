// Small HBM-bound kernels around the conv path: direct (strided / tiny-C_in) Conv1d, the
// ConvTranspose1d polyphase interleave, AdaIN + LeakyReLU + depthwise up-pool, and the
// per-token / per-vector glue used by the denoiser and the ADPM2 sampler.
#include "st2_common.h"

namespace {

// thread = one output (b, co, l); lanes run along l (coalesced stores), co is workgroup-uniform
// so weight reads are scalar broadcasts.
__global__ __launch_bounds__(256) void conv1d_direct_kernel(const float* __restrict__ x, int64_t x_bs, int x_cs,
                                                            const float* __restrict__ w,
                                                            const float* __restrict__ bias, float* __restrict__ y,
                                                            int64_t y_bs, int y_cs, int C_in, int L_in, int L_out,
                                                            int ks, int stride, int pad) {
  const int l = blockIdx.x * 256 + threadIdx.x;
  const int co = blockIdx.y;
  const int b = blockIdx.z;
  if (l >= L_out) return;
  const float* xb = x + (int64_t)b * x_bs;
  const float* wc = w + (int64_t)co * C_in * ks;
  float acc = 0.f;
  const int base = l * stride - pad;
  for (int ci = 0; ci < C_in; ++ci) {
    const float* xr = xb + (int64_t)ci * x_cs;
    for (int t = 0; t < ks; ++t) {
      const int p = base + t;
      if (p >= 0 && p < L_in) acc = fmaf(wc[ci * ks + t], xr[p], acc);
    }
  }
  if (bias) acc += bias[co];
  y[(int64_t)b * y_bs + (int64_t)co * y_cs + l] = acc;
}

// Polyphase split of a strided conv's input: xp[b][ci*stride + r][u] = x[b][ci][u*stride + r - pad] (0 outside).
// Lanes run along the INPUT position p (coalesced reads); a wave's 64 positions scatter to `stride` output rows in
// runs of 64/stride consecutive u.
__global__ __launch_bounds__(256) void phase_split_kernel(const float* __restrict__ x, int64_t x_bs, int x_cs, int L_in,
                                                          float* __restrict__ xp, int64_t p_bs, int p_cs, int Lu,
                                                          int stride, int pad) {
  const int pp = blockIdx.x * 256 + threadIdx.x;  // shifted position p + pad = u*stride + r, in [0, Lu*stride)
  const int ci = blockIdx.y;
  const int b = blockIdx.z;
  if (pp >= Lu * stride) return;
  const int u = pp / stride;
  const int r = pp - u * stride;
  const int p = pp - pad;
  const float v = (p >= 0 && p < L_in) ? x[(int64_t)b * x_bs + (int64_t)ci * x_cs + p] : 0.f;
  xp[(int64_t)b * p_bs + (int64_t)(ci * stride + r) * p_cs + u] = v;
}

// Tile of CVT_TILE outputs per workgroup: the `stride` phase rows it needs are read coalesced (each a contiguous run of
// ~CVT_TILE/stride floats) into LDS and de-interleaved from there, instead of a 6-way gather of 44-byte runs per
// wave; the per-tile (sum, sum of squares) of the stored row segment is emitted for the InstanceNorm that follows
// (fixed-order wave + LDS reduction), so the output is not read again just to be reduced.
constexpr int CVT_TILE = 1024;
// S = the stride as a compile-time constant (2, 3, 5, 6, 10: the up-sampling rates of the two vocoders; 0 = generic, run-time
// divisions).  Round 6: the run-time form spent ~25 VALU slots per output on two integer divisions and moved 4 bytes per lane and
// instruction -- 0.6-0.8 ms per launch where its 12 bytes per output are worth 0.07-0.27 ms at the HBM roof (3.3 ms of a 115 ms
// HiFi-GAN step, 1.2 of the 65 ms default step).  Now every thread owns FOUR CONSECUTIVE outputs (one 16-byte store, one 16-byte load
// of the added tensor when the rows are aligned) and the divisions are by constants.
template <int S>
__global__ __launch_bounds__(256) void convt_interleave_kernel(const float* __restrict__ ph, int64_t p_bs, int p_cs,
                                                               int Lq, const float* __restrict__ bias,
                                                               const float* __restrict__ add, int64_t a_bs, int a_cs,
                                                               float* __restrict__ out, int64_t o_bs, int o_cs, int C,
                                                               int stride_rt, int pad, int L_raw, int reflect_left,
                                                               float* __restrict__ part, int part_nt) {
  extern __shared__ float cvt_tile[];  // [stride][nqp]
  __shared__ float red[2][4];
  const int stride = S ? S : stride_rt;
  const int co = blockIdx.y;
  const int b = blockIdx.z;
  const int L_out = L_raw + reflect_left;
  const int o0 = blockIdx.x * CVT_TILE;
  const int nq = CVT_TILE / stride + 3;
  const int nqp = nq | 1;  // odd row pitch: the de-interleaving reads spread over the banks
  // raw positions this tile touches: l in [l_lo, l_lo + CVT_TILE] (one extra on the left for the reflected sample)
  const int l_lo = max(o0 - reflect_left, 0);
  const int q_lo = (l_lo + pad) / stride;
  // stride x nq staged values, flattened; the loads of 8 iterations are issued together (unconditional, clamped column)
  // before the first LDS store: one load -> one store per iteration serialised `stride` L2 / HBM round trips per tile
  {
    const float* pb = ph + (int64_t)b * p_bs + (int64_t)co * p_cs;
    const int total = stride * nq;
    constexpr int SB = 8;
    for (int e0 = threadIdx.x; e0 < total; e0 += 256 * SB) {
      float t[SB];
#pragma unroll
      for (int u = 0; u < SB; ++u) {
        const int e = min(e0 + u * 256, total - 1);
        const int r = e / nq, i = e - r * nq;
        t[u] = pb[(int64_t)r * C * p_cs + min(q_lo + i, Lq - 1)];
      }
#pragma unroll
      for (int u = 0; u < SB; ++u) {
        const int e = e0 + u * 256;
        if (e < total) {
          const int r = e / nq, i = e - r * nq;
          cvt_tile[r * nqp + i] = (q_lo + i) < Lq ? t[u] : 0.f;
        }
      }
    }
  }
  __syncthreads();
  const float bj = bias ? bias[co] : 0.f;
  const float* ab = add ? add + (int64_t)b * a_bs + (int64_t)co * a_cs : nullptr;
  float* ob = out + (int64_t)b * o_bs + (int64_t)co * o_cs;
  auto staged_at = [&](int o) __attribute__((always_inline)) -> float {  // conv-transpose value of output o, + bias
    int l = o;
    if (reflect_left) l = (o == 0) ? 1 : o - 1;
    const int lp = l + pad;
    const int q = lp / stride;
    const int r = lp - q * stride;
    return cvt_tile[r * nqp + (q - q_lo)] + bj;
  };
  // partial sums are taken of (v - shift), shift = the tile's first stored value (every thread recomputes it: one LDS + one
  // global read): stored behind the sums for st2_stats_finalize -- see st2_conv_epilogue.h on why unshifted sums are not enough
  float shift = staged_at(o0);  // o0 < L_out for every launched tile
  if (ab) shift += ab[o0];
  float s1 = 0.f, s2 = 0.f;
  const int o = o0 + 4 * threadIdx.x;  // this thread's four consecutive outputs
  if (o < L_out) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const bool full = o + 3 < L_out;
    const bool vec = full && ((reinterpret_cast<uintptr_t>(ob + o) & 15) == 0) && (!ab || (reinterpret_cast<uintptr_t>(ab + o) & 15) == 0);
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (full || o + e < L_out) ? staged_at(o + e) : 0.f;
    if (ab) {
      if (vec) {
        const f4 a4 = *reinterpret_cast<const f4*>(ab + o);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += a4[e];
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (full || o + e < L_out) v[e] += ab[o + e];
      }
    }
    if (vec) {
      *reinterpret_cast<f4*>(ob + o) = f4{v[0], v[1], v[2], v[3]};
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (full || o + e < L_out) ob[o + e] = v[e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (full || o + e < L_out) {
        const float dv = v[e] - shift;
        s1 += dv;
        s2 += dv * dv;
      }
  }
  if (part) {
    s1 = st2_wave_sum(s1);
    s2 = st2_wave_sum(s2);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
      red[0][wave] = s1;
      red[1][wave] = s2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      const float t1 = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
      const float t2 = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
      const int64_t slot = ((int64_t)b * C + co) * part_nt + blockIdx.x;
      reinterpret_cast<float2*>(part)[slot] = make_float2(t1, t2);
      part[(int64_t)gridDim.z * C * part_nt * 2 + slot] = shift;  // [B * C][part_nt] shifts behind the sums (st2_stats_finalize)
    }
  }
}

__global__ __launch_bounds__(256) void adain_leaky_pool_kernel(const float* __restrict__ x, int64_t x_bs, int x_cs,
                                                               const float* __restrict__ stats,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, int64_t gb_bs,
                                                               float slope, const float* __restrict__ w,
                                                               const float* __restrict__ bias, float* __restrict__ y,
                                                               int64_t y_bs, int y_cs, int C, int L) {
  const int lo = blockIdx.x * 256 + threadIdx.x;
  const int c = blockIdx.y;
  const int b = blockIdx.z;
  if (lo >= 2 * L) return;
  const float mean = stats[((int64_t)b * C + c) * 2 + 0];
  const float rstd = stats[((int64_t)b * C + c) * 2 + 1];
  const float g = 1.0f + gamma[(int64_t)b * gb_bs + c];
  const float be = beta[(int64_t)b * gb_bs + c];
  const float* xr = x + (int64_t)b * x_bs + (int64_t)c * x_cs;
  auto act = [&](int i) -> float {
    float u = (xr[i] - mean) * rstd;
    u = g * u + be;
    return u >= 0.f ? u : u * slope;
  };
  const int j = lo >> 1;
  float v;
  if ((lo & 1) == 0) {
    v = act(j) * w[c * 3 + 1];
  } else {
    v = act(j) * w[c * 3 + 2];
    if (j + 1 < L) v += act(j + 1) * w[c * 3 + 0];
  }
  if (bias) v += bias[c];
  y[(int64_t)b * y_bs + (int64_t)c * y_cs + lo] = v;
}

__global__ __launch_bounds__(256) void add_chanvec_kernel(const float* __restrict__ x, int64_t x_bs, int x_cs,
                                                          const float* __restrict__ v, int64_t v_bs,
                                                          float* __restrict__ y, int64_t y_bs, int y_cs, int C,
                                                          int N) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  const int c = blockIdx.y;
  const int b = blockIdx.z;
  if (n >= N) return;
  y[(int64_t)b * y_bs + (int64_t)c * y_cs + n] =
      x[(int64_t)b * x_bs + (int64_t)c * x_cs + n] + v[(int64_t)b * v_bs + c];
}

// one wave per (b, c) row
__global__ __launch_bounds__(64) void mean_tokens_kernel(const float* __restrict__ x, int64_t x_bs, int x_cs,
                                                         float* __restrict__ m, int64_t m_bs, int C, int N,
                                                         const int* __restrict__ len) {
  const int c = blockIdx.x;
  const int b = blockIdx.y;
  const float* xr = x + (int64_t)b * x_bs + (int64_t)c * x_cs;
  const int n_b = len ? min(max(len[b], 1), N) : N;  // right-padded batch: tokens >= len[b] do not exist
  double s = 0.0;
  for (int n = threadIdx.x; n < n_b; n += 64) s += (double)xr[n];
  s = st2_wave_sum(s);
  if (threadIdx.x == 0) m[(int64_t)b * m_bs + c] = (float)(s / (double)n_b);
}

__global__ __launch_bounds__(256) void axpbypcz_kernel(const float* __restrict__ x, float a,
                                                       const float* __restrict__ y, float b,
                                                       const float* __restrict__ z, float c, float* __restrict__ out,
                                                       int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float v = a * x[i];
  if (y) v += b * y[i];
  if (z) v += c * z[i];
  out[i] = v;
}

}  // namespace

extern "C" int st2_conv1d_direct(const float* x, int64_t x_bs, int32_t x_cs, const float* w, const float* bias,
                                 float* y, int64_t y_bs, int32_t y_cs, int32_t B, int32_t C_in, int32_t C_out,
                                 int32_t L_in, int32_t L_out, int32_t ks, int32_t stride, int32_t pad,
                                 void* stream) {
  ST2_REQUIRE(x && w && y && B > 0 && C_in > 0 && C_out > 0 && L_in > 0 && L_out > 0 && ks > 0 && stride > 0,
              "st2_conv1d_direct: bad arguments");
  ST2_REQUIRE(C_out <= 65535 && B <= 65535, "st2_conv1d_direct: grid too large");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(conv1d_direct_kernel, dim3(st2_cdiv(L_out, 256), C_out, B), dim3(256), 0, s, x, x_bs, x_cs, w,
                     bias, y, y_bs, y_cs, C_in, L_in, L_out, ks, stride, pad);
  ST2_CHECK_LAUNCH("st2_conv1d_direct");
  return 0;
}

extern "C" int st2_phase_split(const float* x, int64_t x_bs, int32_t x_cs, int32_t B, int32_t C, int32_t L_in,
                               int32_t stride, int32_t pad, float* xp, int64_t p_bs, int32_t p_cs, int32_t Lu,
                               void* stream) {
  ST2_REQUIRE(x && xp && B > 0 && C > 0 && L_in > 0 && stride > 0 && pad >= 0 && Lu > 0,
              "st2_phase_split: bad arguments");
  ST2_REQUIRE(C <= 65535 && B <= 65535 && (int64_t)Lu * stride < (int64_t)1 << 31, "st2_phase_split: grid too large");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(phase_split_kernel, dim3(st2_cdiv((int64_t)Lu * stride, 256), C, B), dim3(256), 0, s, x, x_bs,
                     x_cs, L_in, xp, p_bs, p_cs, Lu, stride, pad);
  ST2_CHECK_LAUNCH("st2_phase_split");
  return 0;
}

extern "C" int st2_convt_interleave(const float* phases, int64_t p_bs, int32_t p_cs, int32_t Lq, const float* bias,
                                    const float* add, int64_t a_bs, int32_t a_cs, float* out, int64_t o_bs,
                                    int32_t o_cs, int32_t B, int32_t C, int32_t stride, int32_t pad, int32_t L_raw,
                                    int32_t reflect_left, void* stream) {
  return st2_convt_interleave_stats(phases, p_bs, p_cs, Lq, bias, add, a_bs, a_cs, out, o_bs, o_cs, B, C, stride, pad,
                                    L_raw, reflect_left, nullptr, 0, stream);
}

extern "C" int st2_convt_interleave_stats(const float* phases, int64_t p_bs, int32_t p_cs, int32_t Lq,
                                          const float* bias, const float* add, int64_t a_bs, int32_t a_cs, float* out,
                                          int64_t o_bs, int32_t o_cs, int32_t B, int32_t C, int32_t stride, int32_t pad,
                                          int32_t L_raw, int32_t reflect_left, float* part, int32_t part_nt,
                                          void* stream) {
  ST2_REQUIRE(phases && out && B > 0 && C > 0 && stride > 0 && L_raw > 0 && Lq > 0,
              "st2_convt_interleave: bad arguments");
  ST2_REQUIRE(reflect_left == 0 || (reflect_left == 1 && L_raw >= 2), "st2_convt_interleave: bad reflect_left");
  ST2_REQUIRE(stride <= 64, "st2_convt_interleave: stride %d too large", stride);
  const int L_out = L_raw + reflect_left;
  const int nt = st2_cdiv(L_out, CVT_TILE);
  if (part) ST2_REQUIRE(part_nt >= nt && (reinterpret_cast<uintptr_t>(part) & 7) == 0,
                        "st2_convt_interleave: part_nt=%d < %d tiles (or part not 8-byte aligned)", part_nt, nt);
  ST2_REQUIRE(B <= 65535 && C <= 65535, "st2_convt_interleave: grid too large");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int nq = CVT_TILE / stride + 3;
  const size_t smem = (size_t)stride * (nq | 1) * sizeof(float);
#define ST2_CVT_LAUNCH(SV)                                                                                              \
  hipLaunchKernelGGL((convt_interleave_kernel<SV>), dim3(nt, C, B), dim3(256), smem, s, phases, p_bs, p_cs, Lq, bias, add, \
                     a_bs, a_cs, out, o_bs, o_cs, C, stride, pad, L_raw, reflect_left, part, part_nt)
  switch (stride) {  // the up-sampling rates of the two vocoders as compile-time constants, anything else generic
    case 2: ST2_CVT_LAUNCH(2); break;
    case 3: ST2_CVT_LAUNCH(3); break;
    case 5: ST2_CVT_LAUNCH(5); break;
    case 6: ST2_CVT_LAUNCH(6); break;
    case 10: ST2_CVT_LAUNCH(10); break;
    default: ST2_CVT_LAUNCH(0); break;
  }
#undef ST2_CVT_LAUNCH
  ST2_CHECK_LAUNCH("st2_convt_interleave");
  return 0;
}

extern "C" int st2_adain_leaky_pool(const float* x, int64_t x_bs, int32_t x_cs, const float* stats,
                                    const float* gamma, const float* beta, int64_t gb_bs, float slope, const float* w,
                                    const float* bias, float* y, int64_t y_bs, int32_t y_cs, int32_t B, int32_t C,
                                    int32_t L, void* stream) {
  ST2_REQUIRE(x && stats && gamma && beta && w && y && B > 0 && C > 0 && L > 0,
              "st2_adain_leaky_pool: bad arguments");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(adain_leaky_pool_kernel, dim3(st2_cdiv(2 * L, 256), C, B), dim3(256), 0, s, x, x_bs, x_cs,
                     stats, gamma, beta, gb_bs, slope, w, bias, y, y_bs, y_cs, C, L);
  ST2_CHECK_LAUNCH("st2_adain_leaky_pool");
  return 0;
}

extern "C" int st2_add_chanvec(const float* x, int64_t x_bs, int32_t x_cs, const float* v, int64_t v_bs, float* y,
                               int64_t y_bs, int32_t y_cs, int32_t B, int32_t C, int32_t N, void* stream) {
  ST2_REQUIRE(x && v && y && B > 0 && C > 0 && N > 0, "st2_add_chanvec: bad arguments");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(add_chanvec_kernel, dim3(st2_cdiv(N, 256), C, B), dim3(256), 0, s, x, x_bs, x_cs, v, v_bs, y,
                     y_bs, y_cs, C, N);
  ST2_CHECK_LAUNCH("st2_add_chanvec");
  return 0;
}

extern "C" int st2_mean_tokens(const float* x, int64_t x_bs, int32_t x_cs, float* m, int64_t m_bs, int32_t B,
                               int32_t C, int32_t N, void* stream) {
  ST2_REQUIRE(x && m && B > 0 && C > 0 && N > 0, "st2_mean_tokens: bad arguments");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(mean_tokens_kernel, dim3(C, B), dim3(64), 0, s, x, x_bs, x_cs, m, m_bs, C, N,
                     static_cast<const int*>(nullptr));
  ST2_CHECK_LAUNCH("st2_mean_tokens");
  return 0;
}

extern "C" int st2_mean_tokens_len(const float* x, int64_t x_bs, int32_t x_cs, float* m, int64_t m_bs, int32_t B,
                                   int32_t C, int32_t N, const int32_t* len, void* stream) {
  ST2_REQUIRE(x && m && B > 0 && C > 0 && N > 0, "st2_mean_tokens_len: bad arguments");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(mean_tokens_kernel, dim3(C, B), dim3(64), 0, s, x, x_bs, x_cs, m, m_bs, C, N,
                     reinterpret_cast<const int*>(len));
  ST2_CHECK_LAUNCH("st2_mean_tokens_len");
  return 0;
}

extern "C" int st2_axpbypcz(const float* x, float a, const float* y, float b, const float* z, float c, float* out,
                            int64_t n, void* stream) {
  ST2_REQUIRE(x && out && n > 0, "st2_axpbypcz: bad arguments");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(axpbypcz_kernel, dim3(st2_cdiv(n, 256)), dim3(256), 0, s, x, a, y, b, z, c, out, n);
  ST2_CHECK_LAUNCH("st2_axpbypcz");
  return 0;
}
