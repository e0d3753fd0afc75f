// Glue kernels between the big ones: everything the launch plans used to do with PyTorch elementwise ops (layout
// changes, broadcasts, the sampler's time features, the duration head and the duration -> alignment expansion), so that
// a module forward is HIP launches only and can be issued from C++ (st2_engine.hip) as well as from Python.
#include "st2_common.h"

namespace {

// four[b] = [t, sin(t*w_j*2*pi) (j < H2), cos(t*w_j*2*pi) (j < H2)], op order of LearnedPositionalEmbedding
// (Modules/diffusion/modules.py:666-671): ((t * w) * 2) * pi in fp32.
__global__ __launch_bounds__(256) void time_features_kernel(float t, const float* __restrict__ w, int H2, int B,
                                                            float* __restrict__ out) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  const int W = 1 + 2 * H2;
  if (j >= W) return;
  float v;
  if (j == 0) {
    v = t;
  } else {
    const int k = (j - 1) % H2;
    const float f = ((t * w[k]) * 2.0f) * 3.14159274101257324f;
    v = (j - 1) < H2 ? sinf(f) : cosf(f);
  }
  out[(int64_t)b * W + j] = v;
}

// y[b][c0 + e][n] = e_in[b][n][e]: token-major embedding -> channel-major rows (32 x 32 tiles through LDS so that both
// sides are coalesced); e_bs = 0 broadcasts one [N][E] table over the batch (the fixed embedding of the CFG branch).
__global__ __launch_bounds__(256) void tokens_to_channels_kernel(const float* __restrict__ e, int64_t e_bs, int N, int E,
                                                                 float* __restrict__ y, int64_t y_bs, int y_cs) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int n0 = blockIdx.x * 32, e0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const float* eb = e + (int64_t)b * e_bs;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int n = n0 + ty + 8 * i, ee = e0 + tx;
    tile[ty + 8 * i][tx] = (n < N && ee < E) ? eb[(int64_t)n * E + ee] : 0.f;
  }
  __syncthreads();
  float* yb = y + (int64_t)b * y_bs;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ee = e0 + ty + 8 * i, n = n0 + tx;
    if (n < N && ee < E) yb[(int64_t)ee * y_cs + n] = tile[tx][ty + 8 * i];
  }
}

// y[b][c][n] = x[b][c] for n < N
__global__ __launch_bounds__(256) void broadcast_cols_kernel(const float* __restrict__ x, int64_t x_bs,
                                                             float* __restrict__ y, int64_t y_bs, int y_cs, int N) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  const int c = blockIdx.y;
  const int b = blockIdx.z;
  if (n >= N) return;
  y[(int64_t)b * y_bs + (int64_t)c * y_cs + n] = x[(int64_t)b * x_bs + c];
}

// y[b][c][l] = x[b][c][l << 0] (strided NCL copy)
__global__ __launch_bounds__(256) void copy_ncl_kernel(const float* __restrict__ x, int64_t x_bs, int x_cs,
                                                       float* __restrict__ y, int64_t y_bs, int y_cs, int L) {
  const int l = blockIdx.x * 256 + threadIdx.x;
  const int c = blockIdx.y;
  const int b = blockIdx.z;
  if (l >= L) return;
  y[(int64_t)b * y_bs + (int64_t)c * y_cs + l] = x[(int64_t)b * x_bs + (int64_t)c * x_cs + l];
}

// Duration head (models.py:450-451 + Demo/Inference_LJSpeech.ipynb:296-301), one wave per token:
//   logits[j] = bias[j] + sum_k W[j][k] * x[b][k][n]  (j < J = max_dur, k < K = 512; fp32, k ascending per lane then a
//   fixed-order wave reduction),  dur = max(1, rint(sum_j sigmoid(logits[j]))),  0 at pad tokens (n >= len[b]),
//   + tail on the utterance's own last token.  x is channel-major [B][K][N] (the duration LSTM's output layout).
__global__ __launch_bounds__(64) void duration_head_kernel(const float* __restrict__ x, int64_t x_bs, int x_cs,
                                                           const float* __restrict__ w, const float* __restrict__ bias,
                                                           int K, int J, int N, const int* __restrict__ len, int tail,
                                                           long long* __restrict__ dur, float* __restrict__ dsum) {
  const int n = blockIdx.x;
  const int b = blockIdx.y;
  const int lane = threadIdx.x;
  const int n_b = len ? min(max(len[b], 1), N) : N;
  const float* xb = x + (int64_t)b * x_bs + n;
  float total = 0.f;
  for (int j = 0; j < J; ++j) {
    float acc = 0.f;
    const float* wj = w + (int64_t)j * K;
    for (int k = lane; k < K; k += 64) acc = fmaf(wj[k], xb[(int64_t)k * x_cs], acc);
    acc = st2_wave_sum(acc);
    acc = __shfl(acc, 0, 64) + bias[j];
    total += 1.0f / (1.0f + expf(-acc));
  }
  if (lane == 0) {
    long long d = (long long)fmaxf(rintf(total), 1.0f);  // torch.round = round-half-even = rintf
    if (n >= n_b) d = 0;
    if (n == n_b - 1) d += tail;
    dur[(int64_t)b * N + n] = d;
    if (dsum) dsum[(int64_t)b * N + n] = total;
  }
}

// Alignment expansion (the one-hot matmul of ipynb:303-312 as a gather): y[b][c][t] = x[b][c][idx(b, t)] with
// idx(b, t) = #{n : cum[b][n] <= t}, cum = inclusive prefix sum of dur[b]; shift = 1 reproduces the HiFi-GAN
// one-frame right shift (y[.., 0] = y_unshifted[.., 0], y[.., t] = y_unshifted[.., t - 1]; Inference_LibriTTS.ipynb).
// Workgroup = (b, 256 frames, 64 channels): the prefix sum of the <= 512 durations is rebuilt in LDS by one wave, every thread
// binary-searches its frame's phoneme once and then copies its column for all channels (reads hit the N-long rows,
// writes are coalesced along t).
__global__ __launch_bounds__(256) void expand_by_durations_kernel(const float* __restrict__ x, int64_t x_bs, int x_cs,
                                                                  const long long* __restrict__ dur, int N, int C,
                                                                  int T, int shift, float* __restrict__ y,
                                                                  int64_t y_bs, int y_cs, int* status) {
  __shared__ int cum[512];
  const int b = blockIdx.y;
  const long long* db = dur + (int64_t)b * N;
  if (threadIdx.x < 64) {  // sequential-in-chunks inclusive scan by one wave (N <= 512)
    int carry = 0;
    for (int base = 0; base < N; base += 64) {
      const int i = base + threadIdx.x;
      int v = i < N ? (int)db[i] : 0;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int u = __shfl_up(v, off, 64);
        if ((int)threadIdx.x >= off) v += u;
      }
      if (i < N) cum[i] = v + carry;
      carry += __shfl(v, 63, 64);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.z == 0 && cum[N - 1] != T)
    st2_raise_status(status, ST2_STATUS_DURATION_SUM);  // the caller's durations do not sum to the frame count it gave
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  const int ts = shift ? max(t - 1, 0) : t;
  int lo = 0, hi = N;  // first n with cum[n] > ts
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (cum[mid] <= ts) lo = mid + 1; else hi = mid;
  }
  const int idx = min(lo, N - 1);
  const float* xb = x + (int64_t)b * x_bs + idx;
  float* yb = y + (int64_t)b * y_bs + t;
  const int c_hi = min(C, (int)(blockIdx.z + 1) * 64);
  for (int c = blockIdx.z * 64; c < c_hi; ++c) yb[(int64_t)c * y_cs] = xb[(int64_t)c * x_cs];
}

// x[b][c][l] = 0 for l >= len[b]  (the masked_fill_ of the text-side modules, models.py:308-312, 547-556)
__global__ __launch_bounds__(256) void mask_tail_kernel(float* __restrict__ x, int64_t x_bs, int x_cs, int L,
                                                        const int* __restrict__ len) {
  const int l = blockIdx.x * 256 + threadIdx.x;
  const int c = blockIdx.y;
  const int b = blockIdx.z;
  if (l >= L || l < len[b]) return;
  x[(int64_t)b * x_bs + (int64_t)c * x_cs + l] = 0.f;
}

// y[b][e][n] = n < len[b] ? (table[tok[b][n]][e] + add[e]) + pos[n][e] : 0: nn.Embedding + transpose + masked_fill of
// TextEncoder.forward (models.py:302-306; add = pos = null) and the word + token-type + position sum of the ALBERT
// embeddings (PL-BERT).  32 tokens x 32 features per tile through LDS: table rows are read along e, y is written along n.
__global__ __launch_bounds__(256) void embed_tokens_kernel(const long long* __restrict__ tok, const float* __restrict__ table,
                                                           int V, int E, int N, const float* __restrict__ add,
                                                           const float* __restrict__ pos, const int* __restrict__ len,
                                                           float* __restrict__ y, int64_t y_bs, int y_cs) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int n0 = blockIdx.x * 32, e0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int nlen = len ? len[b] : N;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int n = n0 + ty + 8 * i, e = e0 + tx;
    float v = 0.f;
    if (n < N && n < nlen && e < E) {
      const long long t = tok[(int64_t)b * N + n];
      if (t >= 0 && t < V) v = table[t * E + e];
      if (add) v += add[e];
      if (pos) v += pos[(int64_t)n * E + e];
    }
    tile[ty + 8 * i][tx] = v;
  }
  __syncthreads();
  float* yb = y + (int64_t)b * y_bs;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int e = e0 + ty + 8 * i, n = n0 + tx;
    if (n < N && e < E) yb[(int64_t)e * y_cs + n] = tile[tx][ty + 8 * i];
  }
}

}  // namespace

extern "C" int st2_embed_tokens(const int64_t* tokens, int32_t B, int32_t N, const float* table, int32_t V, int32_t E,
                                const float* add, const float* pos, const int32_t* len, float* y, int64_t y_bs,
                                int32_t y_cs, void* stream) {
  ST2_REQUIRE(tokens && table && y && B > 0 && N > 0 && V > 0 && E > 0, "st2_embed_tokens: bad arguments");
  ST2_REQUIRE(B <= 65535, "st2_embed_tokens: grid too large");
  hipLaunchKernelGGL(embed_tokens_kernel, dim3(st2_cdiv(N, 32), st2_cdiv(E, 32), B), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const long long*>(tokens), table, V, E, N, add,
                     pos, len, y, y_bs, y_cs);
  ST2_CHECK_LAUNCH("st2_embed_tokens");
  return 0;
}

extern "C" int st2_mask_tail(float* x, int64_t x_bs, int32_t x_cs, int32_t B, int32_t C, int32_t L, const int32_t* len,
                             void* stream) {
  ST2_REQUIRE(x && len && B > 0 && C > 0 && L > 0, "st2_mask_tail: bad arguments");
  ST2_REQUIRE(B <= 65535 && C <= 65535, "st2_mask_tail: grid too large");
  hipLaunchKernelGGL(mask_tail_kernel, dim3(st2_cdiv(L, 256), C, B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x,
                     x_bs, x_cs, L, len);
  ST2_CHECK_LAUNCH("st2_mask_tail");
  return 0;
}

extern "C" int st2_time_features(float t, const float* w, int32_t H2, int32_t B, float* out, void* stream) {
  ST2_REQUIRE(w && out && H2 > 0 && B > 0, "st2_time_features: bad arguments");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(time_features_kernel, dim3(st2_cdiv(1 + 2 * H2, 256), B), dim3(256), 0, s, t, w, H2, B, out);
  ST2_CHECK_LAUNCH("st2_time_features");
  return 0;
}

extern "C" int st2_tokens_to_channels(const float* e, int64_t e_bs, int32_t B, int32_t N, int32_t E, float* y,
                                      int64_t y_bs, int32_t y_cs, void* stream) {
  ST2_REQUIRE(e && y && B > 0 && N > 0 && E > 0, "st2_tokens_to_channels: bad arguments");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(tokens_to_channels_kernel, dim3(st2_cdiv(N, 32), st2_cdiv(E, 32), B), dim3(256), 0, s, e, e_bs, N,
                     E, y, y_bs, y_cs);
  ST2_CHECK_LAUNCH("st2_tokens_to_channels");
  return 0;
}

extern "C" int st2_broadcast_cols(const float* x, int64_t x_bs, float* y, int64_t y_bs, int32_t y_cs, int32_t B,
                                  int32_t C, int32_t N, void* stream) {
  ST2_REQUIRE(x && y && B > 0 && C > 0 && N > 0, "st2_broadcast_cols: bad arguments");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(broadcast_cols_kernel, dim3(st2_cdiv(N, 256), C, B), dim3(256), 0, s, x, x_bs, y, y_bs, y_cs, N);
  ST2_CHECK_LAUNCH("st2_broadcast_cols");
  return 0;
}

extern "C" int st2_copy_ncl(const float* x, int64_t x_bs, int32_t x_cs, float* y, int64_t y_bs, int32_t y_cs, int32_t B,
                            int32_t C, int32_t L, void* stream) {
  ST2_REQUIRE(x && y && B > 0 && C > 0 && L > 0, "st2_copy_ncl: bad arguments");
  ST2_REQUIRE(C <= 65535 && B <= 65535, "st2_copy_ncl: grid too large");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(copy_ncl_kernel, dim3(st2_cdiv(L, 256), C, B), dim3(256), 0, s, x, x_bs, x_cs, y, y_bs, y_cs, L);
  ST2_CHECK_LAUNCH("st2_copy_ncl");
  return 0;
}

extern "C" int st2_duration_head(const float* x, int64_t x_bs, int32_t x_cs, const float* w, const float* bias,
                                 int32_t B, int32_t K, int32_t J, int32_t N, const int32_t* len, int32_t tail,
                                 int64_t* dur, float* dsum, void* stream) {
  ST2_REQUIRE(x && w && bias && dur && B > 0 && K > 0 && J > 0 && N > 0, "st2_duration_head: bad arguments");
  ST2_REQUIRE(B <= 65535, "st2_duration_head: grid too large");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(duration_head_kernel, dim3(N, B), dim3(64), 0, s, x, x_bs, x_cs, w, bias, K, J, N,
                     reinterpret_cast<const int*>(len), tail, reinterpret_cast<long long*>(dur), dsum);
  ST2_CHECK_LAUNCH("st2_duration_head");
  return 0;
}

extern "C" int st2_expand_by_durations(const float* x, int64_t x_bs, int32_t x_cs, const int64_t* dur, int32_t B,
                                       int32_t C, int32_t N, int32_t T, int32_t shift, float* y, int64_t y_bs,
                                       int32_t y_cs, void* stream) {
  ST2_REQUIRE(x && dur && y && B > 0 && C > 0 && N > 0 && T > 0, "st2_expand_by_durations: bad arguments");
  ST2_REQUIRE(N <= 512, "st2_expand_by_durations: N=%d tokens exceed the 512 of PL-BERT's position table", N);
  ST2_REQUIRE(B <= 65535, "st2_expand_by_durations: grid too large");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(expand_by_durations_kernel, dim3(st2_cdiv(T, 256), B, st2_cdiv(C, 64)), dim3(256), 0, s, x, x_bs, x_cs,
                     reinterpret_cast<const long long*>(dur), N, C, T, shift, y, y_bs, y_cs, st2_status_device_ptr());
  ST2_CHECK_LAUNCH("st2_expand_by_durations");
  return 0;
}
