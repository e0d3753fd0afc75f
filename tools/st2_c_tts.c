/*
 * Text -> waveform in plain C: phoneme ids in, audio out, with nothing but include/st2.h + the HIP runtime -- no Python,
 * no PyTorch.  The notebooks' `inference` cell (Demo/Inference_LJSpeech.ipynb:268-315, Demo/Inference_LibriTTS.ipynb:
 * 258-325) as three C-ABI calls per utterance batch:
 *
 *   st2_front_forward     tokens -> t_en, d, s, ref, durations      (text encoder, PL-BERT, style diffusion, duration head)
 *   (host reads the durations: the path's one data-dependent read-back -- the frame count sizes the rest)
 *   st2_prosody_forward   alignment expansion + F0 / energy curves
 *   st2_decoder_forward   vocoder
 *
 * tests/test_c_host.py builds it with gcc and compares its durations and waveform with the Python binding's on the same
 * weights, tokens and noise (bitwise equal: both drive the same C++ launch plans).
 *
 *   st2_c_tts <bundle.bin> <out.bin>
 *
 * bundle.bin (little endian, written by tests/test_c_host.py):
 *   st2_model_config                      raw struct
 *   int32 B, N, steps, tail, shift, has_ref, T_max, has_table;  double embedding_scale, alpha, beta
 *   double sigma0, table[steps-1][11]     (if has_table: the sampler scalars in the caller's arithmetic -- the test passes
 *                                          the Python binding's so that both hosts use identical ones; else st2_sampler_table)
 *   int32 n_weights, then per weight: int32 name_len, name bytes, int32 ndim, int64 shape[ndim], float data[]
 *   int64 tokens[B][N];  float noise[B][2*style], step_noise[steps-1][B][2*style], ref_s[B][2*style] (if has_ref),
 *   float sine_noise[B][600*T_max][9]     (utterance b uses its first 600*T_b rows)
 * out.bin: int64 durations[B][N], then per utterance float wave[600*T_b]
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "st2.h"

#define CHECK_HIP(x)                                                              \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                     \
      return 1;                                                                   \
    }                                                                             \
  } while (0)
#define CHECK_ST2(x)                                                              \
  do {                                                                            \
    if ((x) != 0) {                                                               \
      fprintf(stderr, "%s failed: %s\n", #x, st2_last_error());                   \
      return 1;                                                                   \
    }                                                                             \
  } while (0)

static int read_all(FILE* f, void* dst, size_t n) { return fread(dst, 1, n, f) == n ? 0 : 1; }

static void* upload(FILE* f, size_t bytes) {
  void* h = malloc(bytes);
  void* d = NULL;
  if (!h || read_all(f, h, bytes)) return NULL;
  if (hipMalloc(&d, bytes) != hipSuccess) return NULL;
  if (hipMemcpy(d, h, bytes, hipMemcpyHostToDevice) != hipSuccess) return NULL;
  free(h);
  return d;
}

static void* dev_alloc(size_t bytes) {
  void* d = NULL;
  return hipMalloc(&d, bytes) == hipSuccess ? d : NULL;
}

int main(int argc, char** argv) {
  if (argc != 3) {
    fprintf(stderr, "usage: %s bundle.bin out.bin\n", argv[0]);
    return 2;
  }
  FILE* f = fopen(argv[1], "rb");
  if (!f) {
    perror(argv[1]);
    return 1;
  }
  if (st2_abi_version() != ST2_ABI_VERSION || st2_sizeof_front_args() != (int)sizeof(st2_front_args)) {
    fprintf(stderr, "ABI mismatch: library %d, header %d\n", st2_abi_version(), ST2_ABI_VERSION);
    return 1;
  }
  st2_model_config cfg;
  int32_t hdr[8], n_weights;
  double scal[3];
  if (read_all(f, &cfg, sizeof(cfg)) || read_all(f, hdr, sizeof(hdr)) || read_all(f, scal, sizeof(scal)))
    return 1;
  const int B = hdr[0], N = hdr[1], steps = hdr[2], tail = hdr[3], shift = hdr[4], has_ref = hdr[5], T_max = hdr[6];
  const int sty = cfg.style_dim, C2 = 2 * sty, Cd = cfg.pred_hidden + sty;
  if (B <= 0 || N <= 0 || steps < 2 || T_max <= 0) return 1;
  double* table = (double*)malloc((size_t)(steps - 1) * ST2_SAMPLER_TABLE_COLS * sizeof(double));
  double sigma0 = 0.0;
  if (hdr[7]) {
    if (read_all(f, &sigma0, 8) || read_all(f, table, (size_t)(steps - 1) * ST2_SAMPLER_TABLE_COLS * 8)) return 1;
  } else {
    /* Karras schedule + ADPM2 scalars of the notebooks' sampler (sigma_min 1e-4, sigma_max 3, rho 9; sigma_data 0.2) */
    CHECK_ST2(st2_sampler_table(steps, 1e-4, 3.0, 9.0, 0.2, table, &sigma0));
  }

  if (read_all(f, &n_weights, 4)) return 1;
  st2_engine* eng = NULL;
  CHECK_ST2(st2_create(&cfg, &eng));
  for (int i = 0; i < n_weights; ++i) {
    int32_t name_len, ndim;
    char name[512];
    int64_t shape[8];
    if (read_all(f, &name_len, 4) || name_len <= 0 || name_len >= (int)sizeof(name) || read_all(f, name, name_len)) return 1;
    name[name_len] = 0;
    if (read_all(f, &ndim, 4) || ndim < 0 || ndim > 8 || read_all(f, shape, 8 * (size_t)ndim)) return 1;
    size_t count = 1;
    for (int k = 0; k < ndim; ++k) count *= (size_t)shape[k];
    float* w = (float*)malloc(count * sizeof(float));
    if (!w || read_all(f, w, count * sizeof(float))) return 1;
    CHECK_ST2(st2_load_weights(eng, name, w, shape, ndim));
    free(w);
  }
  CHECK_ST2(st2_finalize_weights(eng, 1 | 2 | 4 | 8 | 16)); /* decoder, denoiser, predictor, text encoder, PL-BERT */

  st2_front_args a;
  memset(&a, 0, sizeof(a));
  a.tokens = (const int64_t*)upload(f, (size_t)B * N * 8);
  a.noise = (const float*)upload(f, (size_t)B * C2 * 4);
  a.step_noise = (const float*)upload(f, (size_t)(steps - 1) * B * C2 * 4);
  a.ref_s = has_ref ? (const float*)upload(f, (size_t)B * C2 * 4) : NULL;
  const size_t noise_rows = (size_t)600 * T_max;
  float* sine = (float*)upload(f, (size_t)B * noise_rows * 9 * 4);
  fclose(f);
  if (!a.tokens || !a.noise || !a.step_noise || (has_ref && !a.ref_s) || !sine) {
    fprintf(stderr, "bundle truncated or device allocation failed\n");
    return 1;
  }
  a.B = B; a.N = N; a.steps = steps; a.tail = tail;
  a.embedding_scale = scal[0]; a.alpha = scal[1]; a.beta = scal[2]; a.t = 0.7;
  a.table = table; a.sigma0 = sigma0;
  a.t_en = (float*)dev_alloc((size_t)B * cfg.dim_in * N * 4);
  a.d_cm = (float*)dev_alloc((size_t)B * Cd * N * 4);
  a.s = (float*)dev_alloc((size_t)B * sty * 4);
  a.ref = (float*)dev_alloc((size_t)B * sty * 4);
  a.durations = (int64_t*)dev_alloc((size_t)B * N * 8);
  const int64_t fw = st2_front_workspace_bytes(eng, &a);
  if (fw <= 0 || !a.t_en || !a.d_cm || !a.s || !a.ref || !a.durations) {
    fprintf(stderr, "st2_front_workspace_bytes / allocation: %s\n", st2_last_error());
    return 1;
  }
  void* ws = dev_alloc((size_t)fw);
  hipStream_t stream;
  CHECK_HIP(hipStreamCreate(&stream));
  CHECK_ST2(st2_front_forward(eng, &a, ws, fw, stream));
  int64_t* dur = (int64_t*)malloc((size_t)B * N * 8);
  CHECK_HIP(hipMemcpyAsync(dur, a.durations, (size_t)B * N * 8, hipMemcpyDeviceToHost, stream));
  CHECK_HIP(hipStreamSynchronize(stream)); /* the one data-dependent read-back: frame counts */
  CHECK_HIP(hipFree(ws));

  FILE* o = fopen(argv[2], "wb");
  if (!o || fwrite(dur, 8, (size_t)B * N, o) != (size_t)B * N) {
    perror(argv[2]);
    return 1;
  }
  for (int b = 0; b < B; ++b) { /* one utterance per decoder call: its InstanceNorms span exactly its own frames */
    int64_t T = 0;
    for (int n = 0; n < N; ++n) T += dur[(size_t)b * N + n];
    if (T <= 0 || T > T_max) {
      fprintf(stderr, "utterance %d: %lld frames (bundle carries noise for %d)\n", b, (long long)T, T_max);
      return 1;
    }
    const size_t L = (size_t)600 * (size_t)T;
    float* asr = (float*)dev_alloc((size_t)cfg.dim_in * T * 4);
    float* f0 = (float*)dev_alloc((size_t)2 * T * 4);
    float* nn = (float*)dev_alloc((size_t)2 * T * 4);
    float* wave = (float*)dev_alloc(L * 4);
    const int64_t pw = st2_prosody_workspace_bytes(eng, 1, N, (int32_t)T), dw = st2_decoder_workspace_bytes(eng, 1, (int32_t)T);
    if (pw <= 0 || dw <= 0 || !asr || !f0 || !nn || !wave) {
      fprintf(stderr, "workspace query / allocation failed: %s\n", st2_last_error());
      return 1;
    }
    void* w2 = dev_alloc((size_t)(pw > dw ? pw : dw));
    CHECK_ST2(st2_prosody_forward(eng, a.d_cm + (size_t)b * Cd * N, a.t_en + (size_t)b * cfg.dim_in * N,
                                  a.durations + (size_t)b * N, a.s + (size_t)b * sty, 1, N, (int32_t)T, shift, asr, f0, nn, w2,
                                  pw, stream));
    CHECK_ST2(st2_decoder_forward(eng, asr, f0, nn, a.ref + (size_t)b * sty, sine + (size_t)b * noise_rows * 9, NULL, 1,
                                  (int32_t)T, wave, w2, dw, NULL, stream));
    float* h = (float*)malloc(L * 4);
    CHECK_HIP(hipMemcpyAsync(h, wave, L * 4, hipMemcpyDeviceToHost, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    if (fwrite(h, 4, L, o) != L) {
      perror(argv[2]);
      return 1;
    }
    free(h);
    CHECK_HIP(hipFree(w2)); CHECK_HIP(hipFree(asr)); CHECK_HIP(hipFree(f0)); CHECK_HIP(hipFree(nn)); CHECK_HIP(hipFree(wave));
    printf("st2_c_tts: utterance %d: %d phonemes -> %lld frames -> %zu samples\n", b, N, (long long)T, L);
  }
  fclose(o);
  if (st2_status(0) != 0) fprintf(stderr, "warning: device status word = %d (see st2.h)\n", st2_status(0));
  CHECK_ST2(st2_destroy(eng));
  return 0;
}
