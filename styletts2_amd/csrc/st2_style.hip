// Reference-audio style path: the pieces of the mel front-end (meldataset.py:58-66 ->
// torchaudio MelSpectrogram) and of StyleEncoder (models.py:139-164) that are not a Conv1d-shaped GEMM.  The GEMM-shaped
// parts -- the windowed DFT (a k=1 conv over frame columns), the mel filter bank, every 3x3 / 5x5 / 1x1 Conv2d (as a
// Conv1d over the width with the kernel rows stacked along the channels, see styletts2_amd/style.py) -- run on the
// split-f16 MFMA conv kernels.  Everything here is HBM-bound elementwise / gather work on small tensors.
//
// 2-D feature maps are stored row-major over (h, c, w): element (b, h, c, w) at x + b*x_bs + h*x_hs + c*x_cs + w, so
// that one image row (all channels) is an NCL tensor [C][W] and three consecutive rows are the 3C-channel input of the
// row-stacked Conv1d.
#include "st2_common.h"

namespace {

// frames[b][c][m] = wave[b][reflect(m*hop + c - shift)], c < n_win, m < M: the columns of torch.stft's frame matrix
// (center=True, pad_mode="reflect") restricted to the n_win taps where the zero-padded window is non-zero
// (shift = n_fft/2 - (n_fft - n_win)/2).
__global__ __launch_bounds__(256) void stft_frames_kernel(const float* __restrict__ wave, int64_t w_bs, int L, int n_win,
                                                          int hop, int shift, int M, float* __restrict__ fr,
                                                          int64_t f_bs, int f_cs) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  const int c = blockIdx.y;
  const int b = blockIdx.z;
  if (m >= M) return;
  int i = m * hop + c - shift;
  if (i < 0) i = -i;
  if (i >= L) i = 2 * (L - 1) - i;
  fr[(int64_t)b * f_bs + (int64_t)c * f_cs + m] = wave[(int64_t)b * w_bs + i];
}

// p[b][k][m] = y[b][k][m]^2 + y[b][K + k][m]^2   (|X_k|^2 from the stacked real / imaginary DFT rows)
__global__ __launch_bounds__(256) void power_spectrum_kernel(const float* __restrict__ y, int64_t y_bs, int y_cs, int K,
                                                             int M, float* __restrict__ p, int64_t p_bs, int p_cs) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  const int k = blockIdx.y;
  const int b = blockIdx.z;
  if (m >= M) return;
  const float re = y[(int64_t)b * y_bs + (int64_t)k * y_cs + m];
  const float im = y[(int64_t)b * y_bs + (int64_t)(K + k) * y_cs + m];
  p[(int64_t)b * p_bs + (int64_t)k * p_cs + m] = re * re + im * im;
}

// x[i] = (log(eps + x[i]) - mean) / std  in place (meldataset.py:63-65)
__global__ __launch_bounds__(256) void log_norm_kernel(float* __restrict__ x, int64_t n, float eps, float mean,
                                                       float stdv) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  x[i] = (logf(eps + x[i]) - mean) / stdv;
}

// Depthwise Conv2d(C, C, 3, stride 2, padding 1, groups = C) -- LearnedDownSample('half'), models.py:27-42:
// y[b][ho][c][wo] = bias[c] + sum_{dh,dw} w[c][dh][dw] * x[b][2ho + dh - 1][c][2wo + dw - 1], zero outside the map.
__global__ __launch_bounds__(256) void dwconv3x3s2_kernel(const float* __restrict__ x, int64_t x_bs, int64_t x_hs,
                                                          int x_cs, const float* __restrict__ w,
                                                          const float* __restrict__ bias, int H, int W, int Ho, int Wo,
                                                          float* __restrict__ y, int64_t y_bs, int64_t y_hs, int y_cs) {
  const int wo = blockIdx.x * 256 + threadIdx.x;
  const int c = blockIdx.y;
  const int b = blockIdx.z / Ho, ho = blockIdx.z % Ho;
  if (wo >= Wo) return;
  const float* wc = w + c * 9;
  float acc = bias ? bias[c] : 0.f;
#pragma unroll
  for (int dh = 0; dh < 3; ++dh) {
    const int h = 2 * ho + dh - 1;
    if (h < 0 || h >= H) continue;
    const float* xr = x + (int64_t)b * x_bs + (int64_t)h * x_hs + (int64_t)c * x_cs;
#pragma unroll
    for (int dw = 0; dw < 3; ++dw) {
      const int ww = 2 * wo + dw - 1;
      if (ww >= 0 && ww < W) acc += wc[dh * 3 + dw] * xr[ww];
    }
  }
  y[(int64_t)b * y_bs + (int64_t)ho * y_hs + (int64_t)c * y_cs + wo] = acc;
}

// DownSample('half'), models.py:72-75: the last column is replicated when the width is odd, then F.avg_pool2d(x, 2).
__global__ __launch_bounds__(256) void avgpool2x2_kernel(const float* __restrict__ x, int64_t x_bs, int64_t x_hs, int x_cs,
                                                         int W, int Ho, int Wo, float* __restrict__ y, int64_t y_bs,
                                                         int64_t y_hs, int y_cs) {
  const int wo = blockIdx.x * 256 + threadIdx.x;
  const int c = blockIdx.y;
  const int b = blockIdx.z / Ho, ho = blockIdx.z % Ho;
  if (wo >= Wo) return;
  const float* r0 = x + (int64_t)b * x_bs + (int64_t)(2 * ho) * x_hs + (int64_t)c * x_cs;
  const float* r1 = r0 + x_hs;
  const int w0 = 2 * wo, w1 = min(2 * wo + 1, W - 1);
  const float s = ((r0[w0] + r0[w1]) + r1[w0]) + r1[w1];
  y[(int64_t)b * y_bs + (int64_t)ho * y_hs + (int64_t)c * y_cs + wo] = s * 0.25f;
}

}  // namespace

extern "C" int st2_stft_frames(const float* wave, int64_t w_bs, int32_t B, int32_t L, int32_t n_win, int32_t hop,
                               int32_t shift, float* frames, int64_t f_bs, int32_t f_cs, void* stream) {
  ST2_REQUIRE(wave && frames && B > 0 && L > 1 && n_win > 0 && hop > 0, "st2_stft_frames: bad arguments");
  // The last frame starts at (L / hop) * hop and reads positions up to i = (L / hop) * hop + n_win - 1 - shift; a single
  // reflection maps i > L - 1 to 2 (L - 1) - i, which must not go negative (and the left one, shift - j, not past L - 1).
  ST2_REQUIRE(shift >= 0 && shift < L && (int64_t)(L / hop) * hop + n_win - 1 - shift <= 2 * (int64_t)(L - 1),
              "st2_stft_frames: reflection reaches past the signal (L=%d, n_win=%d, hop=%d, shift=%d)", L, n_win, hop, shift);
  const int M = L / hop + 1;
  dim3 grid(st2_cdiv(M, 256), n_win, B);
  hipLaunchKernelGGL(stft_frames_kernel, grid, dim3(256), 0, (hipStream_t)stream, wave, w_bs, L, n_win, hop, shift, M,
                     frames, f_bs, f_cs);
  ST2_CHECK_LAUNCH("st2_stft_frames");
  return 0;
}

extern "C" int st2_power_spectrum(const float* y, int64_t y_bs, int32_t y_cs, int32_t B, int32_t K, int32_t M, float* p,
                                  int64_t p_bs, int32_t p_cs, void* stream) {
  ST2_REQUIRE(y && p && B > 0 && K > 0 && M > 0, "st2_power_spectrum: bad arguments");
  dim3 grid(st2_cdiv(M, 256), K, B);
  hipLaunchKernelGGL(power_spectrum_kernel, grid, dim3(256), 0, (hipStream_t)stream, y, y_bs, y_cs, K, M, p, p_bs, p_cs);
  ST2_CHECK_LAUNCH("st2_power_spectrum");
  return 0;
}

extern "C" int st2_log_norm(float* x, int64_t n, float eps, float mean, float stdv, void* stream) {
  ST2_REQUIRE(x && n > 0 && stdv != 0.f, "st2_log_norm: bad arguments");
  hipLaunchKernelGGL(log_norm_kernel, dim3(st2_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, n, eps, mean, stdv);
  ST2_CHECK_LAUNCH("st2_log_norm");
  return 0;
}

extern "C" int st2_dwconv3x3s2(const float* x, int64_t x_bs, int64_t x_hs, int32_t x_cs, const float* w,
                               const float* bias, int32_t B, int32_t C, int32_t H, int32_t W, float* y, int64_t y_bs,
                               int64_t y_hs, int32_t y_cs, void* stream) {
  ST2_REQUIRE(x && w && y && B > 0 && C > 0 && H > 0 && W > 0, "st2_dwconv3x3s2: bad arguments");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  ST2_REQUIRE((int64_t)B * Ho <= 65535 && C <= 65535, "st2_dwconv3x3s2: grid too large");
  dim3 grid(st2_cdiv(Wo, 256), C, B * Ho);
  hipLaunchKernelGGL(dwconv3x3s2_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, x_bs, x_hs, x_cs, w, bias, H, W, Ho,
                     Wo, y, y_bs, y_hs, y_cs);
  ST2_CHECK_LAUNCH("st2_dwconv3x3s2");
  return 0;
}

extern "C" int st2_avgpool2x2(const float* x, int64_t x_bs, int64_t x_hs, int32_t x_cs, int32_t B, int32_t C, int32_t H,
                              int32_t W, float* y, int64_t y_bs, int64_t y_hs, int32_t y_cs, void* stream) {
  ST2_REQUIRE(x && y && B > 0 && C > 0 && H > 1 && W > 0, "st2_avgpool2x2: bad arguments");
  ST2_REQUIRE(H % 2 == 0, "st2_avgpool2x2: odd height %d (the reference pads the width only, models.py:72-75)", H);
  const int Ho = H / 2, Wo = (W + 1) / 2;
  ST2_REQUIRE((int64_t)B * Ho <= 65535 && C <= 65535, "st2_avgpool2x2: grid too large");
  dim3 grid(st2_cdiv(Wo, 256), C, B * Ho);
  hipLaunchKernelGGL(avgpool2x2_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, x_bs, x_hs, x_cs, W, Ho, Wo, y, y_bs,
                     y_hs, y_cs);
  ST2_CHECK_LAUNCH("st2_avgpool2x2");
  return 0;
}
