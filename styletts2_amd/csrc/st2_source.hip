// Harmonic-plus-noise source (SineGen / SourceModuleHnNSF), its STFT features, and the iSTFT
// synthesis head of the iSTFTNet vocoder.  All three are HBM/latency-bound sample-rate kernels.
//
// The SineGen phase path reproduces the ATen-CPU op order bit for bit:
// phases reach ~9e4 rad where one fp32 ulp is 7.8e-3 rad, so any re-association shows up as an
// O(1e-2) difference in sin().  The library is built with -ffp-contract=off; fused ops appear only
// where ATen itself fuses (the linear-interpolation fmaf pair).
#include "st2_common.h"

namespace {

__device__ __forceinline__ float torch_remainder1(float x) {
  // torch `x % 1` (ATen remainder): fmod, then shift into [0, 1) when the signs differ.
  float m = fmodf(x, 1.0f);
  if (m != 0.0f && m < 0.0f) m += 1.0f;
  return m;
}

// ---- pass 1: frame-rate phase  phase_f[b][h][k] = ((cumsum_k rad) * 2) * pi_f32 * U ------------
// The prefix sums are taken in fp64 and each prefix is rounded to fp32, as ATen-CPU cumsum does (istftnet.py:183-184).
// One WAVE per (b, h) row: lane i sums its contiguous chunk of the row, the chunk totals are scanned across the wave,
// then every lane re-walks its chunk from its offset.  (Round 2: one THREAD per row walking 800 frames with a load, a
// dependent fp64 add and a store per iteration -- 250 us of pure latency per decoder call.)  The fp64 sums are exact to
// ~1e-13 relative whatever the association (<= 2^10 terms of 24-bit values), far below the fp32 rounding of a prefix, so the
// stored phases are those of the sequential scan.
__global__ __launch_bounds__(64) void sinegen_phase_kernel(const float* __restrict__ f0, int B, int F, int U, int H,
                                                           float sample_rate, float* __restrict__ phase_f) {
  const int row = blockIdx.x;  // (b, h)
  const int lane = threadIdx.x;
  const int b = row / H;
  const int h = row % H;
  const float mult = (float)(h + 1);
  const float* f = f0 + (int64_t)b * F;
  float* out = phase_f + ((int64_t)b * H + h) * F;
  const float two_pi_part = 3.14159274101257324219f;  // (float)np.pi
  const float fu = (float)U;
  const int chunk = (F + 63) / 64;
  const int k0 = min(lane * chunk, F), k1 = min(k0 + chunk, F);
  double part = 0.0;
  for (int k = k0; k < k1; ++k) {
    const float fn = f[k] * mult;                          // istftnet.py:228
    part += (double)torch_remainder1(fn / sample_rate);    // istftnet.py:152
  }
  // inclusive scan of the chunk totals across the wave (fixed order: Hillis-Steele over 64 lanes), then exclusive offset
  double incl = part;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const double up = __shfl_up(incl, d, 64);
    if (lane >= d) incl += up;
  }
  double acc = __shfl_up(incl, 1, 64);  // exclusive offset = the inclusive total of the lane below
  if (lane == 0) acc = 0.0;
  for (int k = k0; k < k1; ++k) {
    const float fn = f[k] * mult;
    acc += (double)torch_remainder1(fn / sample_rate);
    const float c = (float)acc;
    out[k] = ((c * 2.0f) * two_pi_part) * fu;
  }
}

// ---- pass 2: sample-rate sines, U/V mix, noise, 9->1 linear, tanh ------------------------------
__global__ __launch_bounds__(256) void har_source_kernel(const float* __restrict__ f0, int F, int U, int H,
                                                         const float* __restrict__ noise,
                                                         const float* __restrict__ lin_w,
                                                         const float* __restrict__ lin_b, float sine_amp,
                                                         float noise_std, float voiced_threshold,
                                                         const float* __restrict__ phase_f,
                                                         float* __restrict__ out) {
  const int L = F * U;
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (t >= L) return;
  // F.interpolate(scale_factor=U, mode='linear', align_corners=False) source index, ATen-CPU form
  const float rs = (float)(1.0 / (double)U);
  float src = fmaf(rs, (float)t + 0.5f, -0.5f);
  src = src < 0.f ? 0.f : src;
  const int i0 = (int)src;
  const int i1 = min(i0 + 1, F - 1);
  const float w1 = src - (float)i0;
  const float w0 = 1.0f - w1;

  const float f0v = f0[(int64_t)b * F + t / U];  // nearest x U up-sampling (istftnet.py:314,352)
  const float uv = f0v > voiced_threshold ? 1.0f : 0.0f;
  const float noise_amp = uv * noise_std + ((1.0f - uv) * sine_amp) / 3.0f;  // istftnet.py:241

  const float* nz = noise + ((int64_t)b * L + t) * H;
  float acc = 0.f;
  for (int h = 0; h < H; ++h) {
    const float* pf = phase_f + ((int64_t)b * H + h) * F;
    const float ph = fmaf(w0, pf[i0], w1 * pf[i1]);
    const float sine = sinf(ph) * sine_amp;
    const float sw = sine * uv + noise_amp * nz[h];
    acc = fmaf(lin_w[h], sw, acc);
  }
  out[(int64_t)b * L + t] = tanhf(acc + lin_b[0]);
}

// ---- STFT: one thread per frame, N-point DFT by table (N <= 32) ---------------------------------
constexpr int MAXN = 32;
__global__ __launch_bounds__(256) void stft_kernel(const float* __restrict__ x, int L, int N, int hop,
                                                   float* __restrict__ har, int64_t har_bs, int har_cs) {
  __shared__ float tw_c[MAXN], tw_s[MAXN], win[MAXN];
  if (threadIdx.x < N) {
    // cospi/sinpi are exact at multiples of 1/2: the DC and Nyquist bins then have an exactly zero
    // imaginary part, as a real FFT gives, so atan2 returns +pi (not a random +-pi) when Re < 0.
    const double a = 2.0 * (double)threadIdx.x / (double)N;
    tw_c[threadIdx.x] = (float)cospi(a);
    tw_s[threadIdx.x] = (float)sinpi(a);
    win[threadIdx.x] = (float)(0.5 - 0.5 * cospi(a));
  }
  __syncthreads();
  const int M = L / hop + 1;
  const int m = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (m >= M) return;
  const float* xb = x + (int64_t)b * L;
  float fr[MAXN];
  const int pad = N / 2;
#pragma unroll
  for (int n = 0; n < MAXN; ++n) {
    if (n < N) {
      int i = m * hop + n - pad;
      if (i < 0) i = -i;
      if (i >= L) i = 2 * (L - 1) - i;
      fr[n] = xb[i] * win[n];
    } else {
      fr[n] = 0.f;
    }
  }
  const int NB = N / 2 + 1;
  float* hb = har + (int64_t)b * har_bs + m;
  for (int k = 0; k < NB; ++k) {
    float re = 0.f, im = 0.f;
    int idx = 0;
#pragma unroll
    for (int n = 0; n < MAXN; ++n) {
      if (n < N) {
        re = fmaf(fr[n], tw_c[idx], re);
        im = fmaf(-fr[n], tw_s[idx], im);
        idx += k;
        if (idx >= N) idx -= N;
      }
    }
    hb[(int64_t)k * har_cs] = sqrtf(re * re + im * im);
    hb[(int64_t)(NB + k) * har_cs] = atan2f(im, re);
  }
}

// ---- iSTFT: one thread per output sample; gathers the <= N/hop frames that overlap it ------------
// The polar -> cartesian conversion (mag * cos(phase), mag * sin(phase)) of the frames a workgroup's 256 samples touch is
// done ONCE per (frame, bin) into LDS ([bin][frame], frames along the lanes): per sample it was repeated for every one
// of the N/hop overlapping frames and N bins -- 20 x the sincos work at n_fft = 20, hop = 5.  Same values, same
// accumulation order: the waveform is bitwise unchanged.
__global__ __launch_bounds__(256) void istft_kernel(const float* __restrict__ sp, int64_t sp_bs, int sp_cs, int M,
                                                    int N, int hop, float* __restrict__ wave, int64_t wave_bs, int FRP) {
  __shared__ float tw_c[MAXN], tw_s[MAXN], win[MAXN];
  extern __shared__ float ri[];  // [2][NB][FRP]: re, im of the frames m_base .. m_base + FR - 1
  if (threadIdx.x < N) {
    // cospi/sinpi are exact at multiples of 1/2: the DC and Nyquist bins then have an exactly zero
    // imaginary part, as a real FFT gives, so atan2 returns +pi (not a random +-pi) when Re < 0.
    const double a = 2.0 * (double)threadIdx.x / (double)N;
    tw_c[threadIdx.x] = (float)cospi(a);
    tw_s[threadIdx.x] = (float)sinpi(a);
    win[threadIdx.x] = (float)(0.5 - 0.5 * cospi(a));
  }
  const int Lw = hop * (M - 1);
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  const int NB = N / 2 + 1;
  const float* sb = sp + (int64_t)b * sp_bs;
  {  // frames touched by this workgroup's samples t0 .. t0 + 255
    const int t0 = blockIdx.x * 256;
    const int tp0 = t0 + N / 2, tp1 = min(t0 + 255, Lw - 1) + N / 2;
    const int m_base = (tp0 - N + 1 <= 0) ? 0 : (tp0 - N + hop) / hop;
    const int m_last = min(tp1 / hop, M - 1);
    const int FR = m_last - m_base + 1;  // <= FRP
    float* re_s = ri;
    float* im_s = ri + NB * FRP;
    for (int e = threadIdx.x; e < NB * FR; e += 256) {
      const int k = e / FR, f = e - k * FR;
      const float mag = sb[(int64_t)k * sp_cs + m_base + f];
      const float ph = sb[(int64_t)(NB + k) * sp_cs + m_base + f];
      re_s[k * FRP + f] = mag * cosf(ph);
      im_s[k * FRP + f] = mag * sinf(ph);
    }
  }
  __syncthreads();
  if (t >= Lw) return;
  const int tp = t + N / 2;  // position in the un-trimmed overlap-add buffer
  const int tp0 = (int)blockIdx.x * 256 + N / 2;  // (int: blockIdx is unsigned, tp0 - N + 1 must be able to go negative)
  const int m_base = (tp0 - N + 1 <= 0) ? 0 : (tp0 - N + hop) / hop;
  const float* re_s = ri;
  const float* im_s = ri + NB * FRP;
  int m_hi = tp / hop;
  if (m_hi > M - 1) m_hi = M - 1;
  int m_lo = (tp - N + hop) / hop;  // ceil((tp - N + 1) / hop) for tp-N+1 >= 0
  if (tp - N + 1 <= 0) m_lo = 0;
  const float inv_n = 1.0f / (float)N;
  float y = 0.f, env = 0.f;
  for (int m = m_lo; m <= m_hi; ++m) {
    const int n = tp - m * hop;  // 0 <= n < N
    // irfft bin sum: (1/N) [ Re S0 + (-1)^n Re S_{N/2} + 2 sum_{k=1}^{N/2-1} (Re S_k cos - Im S_k sin) ]
    float acc = 0.f;
    int idx = 0;
    const int f = m - m_base;
    for (int k = 0; k < NB; ++k) {
      const float re = re_s[k * FRP + f];
      const float im = im_s[k * FRP + f];
      if (k == 0 || k == N / 2) {
        acc += re * tw_c[idx];  // imaginary parts of DC / Nyquist are ignored (c2r semantics)
      } else {
        acc += 2.0f * (re * tw_c[idx] - im * tw_s[idx]);
      }
      idx += n;
      if (idx >= N) idx -= N;
    }
    const float w = win[n];
    y += (acc * inv_n) * w;
    env += w * w;
  }
  wave[(int64_t)b * wave_bs + t] = y / env;
}

}  // namespace

extern "C" int st2_har_source(const float* f0, int32_t B, int32_t F, int32_t U, int32_t H, const float* noise,
                              const float* lin_w, const float* lin_b, float sine_amp, float noise_std,
                              float voiced_threshold, float sample_rate, float* phase_scratch, float* out,
                              void* stream) {
  ST2_REQUIRE(f0 && noise && lin_w && lin_b && phase_scratch && out, "st2_har_source: null pointer");
  ST2_REQUIRE(B > 0 && F > 0 && U > 0 && H > 0 && H <= 64, "st2_har_source: bad geometry");
  ST2_REQUIRE((int64_t)F * U < (1LL << 31), "st2_har_source: utterance too long");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(sinegen_phase_kernel, dim3(B * H), dim3(64), 0, s, f0, B, F, U, H, sample_rate,
                     phase_scratch);
  ST2_CHECK_LAUNCH("st2_har_source(phase)");
  hipLaunchKernelGGL(har_source_kernel, dim3(st2_cdiv((int64_t)F * U, 256), B), dim3(256), 0, s, f0, F, U, H, noise,
                     lin_w, lin_b, sine_amp, noise_std, voiced_threshold, phase_scratch, out);
  ST2_CHECK_LAUNCH("st2_har_source");
  return 0;
}

extern "C" int st2_stft_mag_phase(const float* x, int32_t B, int32_t L, int32_t n_fft, int32_t hop, float* har,
                                  int64_t har_bs, int32_t har_cs, void* stream) {
  ST2_REQUIRE(x && har && B > 0 && L > 0, "st2_stft_mag_phase: bad arguments");
  ST2_REQUIRE(n_fft >= 2 && n_fft <= MAXN && (n_fft % 2) == 0 && hop > 0 && L > n_fft / 2,
              "st2_stft_mag_phase: n_fft=%d hop=%d L=%d unsupported", n_fft, hop, L);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int M = L / hop + 1;
  hipLaunchKernelGGL(stft_kernel, dim3(st2_cdiv(M, 256), B), dim3(256), 0, s, x, L, n_fft, hop, har, har_bs, har_cs);
  ST2_CHECK_LAUNCH("st2_stft_mag_phase");
  return 0;
}

extern "C" int st2_istft(const float* sp, int64_t sp_bs, int32_t sp_cs, int32_t B, int32_t M, int32_t n_fft,
                         int32_t hop, float* wave, int64_t wave_bs, void* stream) {
  ST2_REQUIRE(sp && wave && B > 0 && M > 1, "st2_istft: bad arguments");
  ST2_REQUIRE(n_fft >= 2 && n_fft <= MAXN && (n_fft % 2) == 0 && hop > 0 && (n_fft % hop) == 0,
              "st2_istft: n_fft=%d hop=%d unsupported", n_fft, hop);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int Lw = hop * (M - 1);
  const int FRP = (256 / hop + n_fft / hop + 3) | 1;  // frames a 256-sample tile can touch, odd pitch
  const size_t smem = (size_t)2 * (n_fft / 2 + 1) * FRP * sizeof(float);
  hipLaunchKernelGGL(istft_kernel, dim3(st2_cdiv(Lw, 256), B), dim3(256), smem, s, sp, sp_bs, sp_cs, M, n_fft, hop, wave,
                     wave_bs, FRP);
  ST2_CHECK_LAUNCH("st2_istft");
  return 0;
}
