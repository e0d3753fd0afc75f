/*
 * A host in plain C: synthesises a batch with nothing but include/st2.h + the HIP runtime -- no Python, no PyTorch.
 * It is what a non-Python deployment of the reference's Decoder.forward (Modules/istftnet.py:499-528 /
 * Modules/hifigan.py:446-475) looks like on this engine, and what tests/test_c_host.py builds (gcc) and runs against
 * the Python binding on the same weights and inputs (bitwise equal).
 *
 *   st2_c_host <bundle.bin> <wave_out.bin>
 *
 * bundle.bin (little endian, written by tests/test_c_host.py):
 *   st2_model_config                      raw struct
 *   int32 B, T
 *   int32 n_weights, then per weight: int32 name_len, name bytes, int32 ndim, int64 shape[ndim], float data[]
 *   float asr[B][dim_in][T], f0[B][2T], n[B][2T], s[B][style_dim], sine_noise[B][600T][9]
 * wave_out.bin: float wave[B][600T]
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

static float* upload(FILE* f, size_t count) {
  float* h = (float*)malloc(count * sizeof(float));
  float* d = NULL;
  if (!h || read_all(f, h, count * sizeof(float))) return NULL;
  if (hipMalloc((void**)&d, count * sizeof(float)) != hipSuccess) return NULL;
  if (hipMemcpy(d, h, count * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return NULL;
  free(h);
  return d;
}

int main(int argc, char** argv) {
  if (argc != 3) {
    fprintf(stderr, "usage: %s bundle.bin wave_out.bin\n", argv[0]);
    return 2;
  }
  FILE* f = fopen(argv[1], "rb");
  if (!f) {
    perror(argv[1]);
    return 1;
  }
  if (st2_abi_version() != ST2_ABI_VERSION) {
    fprintf(stderr, "ABI mismatch: library %d, header %d\n", st2_abi_version(), ST2_ABI_VERSION);
    return 1;
  }
  st2_model_config cfg;
  int32_t B, T, n_weights;
  if (read_all(f, &cfg, sizeof(cfg)) || read_all(f, &B, 4) || read_all(f, &T, 4) || read_all(f, &n_weights, 4)) return 1;

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
    CHECK_ST2(st2_load_weights(eng, name, w, shape, ndim)); /* copied: the host buffer is ours again */
    free(w);
  }
  CHECK_ST2(st2_finalize_weights(eng, 1)); /* 1 = decoder: packs and uploads in one device allocation */

  const size_t L = (size_t)600 * T;
  float* asr = upload(f, (size_t)B * cfg.dim_in * T);
  float* f0 = upload(f, (size_t)B * 2 * T);
  float* nn = upload(f, (size_t)B * 2 * T);
  float* s = upload(f, (size_t)B * cfg.style_dim);
  float* noise = upload(f, (size_t)B * L * 9);
  fclose(f);
  if (!asr || !f0 || !nn || !s || !noise) {
    fprintf(stderr, "bundle truncated or device allocation failed\n");
    return 1;
  }
  float* wave = NULL;
  void* ws = NULL;
  const int64_t ws_bytes = st2_decoder_workspace_bytes(eng, B, T);
  if (ws_bytes <= 0) {
    fprintf(stderr, "st2_decoder_workspace_bytes: %s\n", st2_last_error());
    return 1;
  }
  CHECK_HIP(hipMalloc((void**)&wave, (size_t)B * L * sizeof(float)));
  CHECK_HIP(hipMalloc(&ws, (size_t)ws_bytes));
  hipStream_t stream;
  CHECK_HIP(hipStreamCreate(&stream));
  /* the call queues the whole Decoder.forward launch plan on `stream` and returns; run it twice: the second call reuses
     the workspace and must reproduce the first bit for bit */
  CHECK_ST2(st2_decoder_forward(eng, asr, f0, nn, s, noise, NULL, B, T, wave, ws, ws_bytes, NULL, stream));
  CHECK_ST2(st2_decoder_forward(eng, asr, f0, nn, s, noise, NULL, B, T, wave, ws, ws_bytes, NULL, stream));
  CHECK_HIP(hipStreamSynchronize(stream));
  if (st2_status(0) != 0) fprintf(stderr, "warning: device status word = %d (see st2.h)\n", st2_status(0));

  float* h = (float*)malloc((size_t)B * L * sizeof(float));
  CHECK_HIP(hipMemcpy(h, wave, (size_t)B * L * sizeof(float), hipMemcpyDeviceToHost));
  FILE* o = fopen(argv[2], "wb");
  if (!o || fwrite(h, sizeof(float), (size_t)B * L, o) != (size_t)B * L) {
    perror(argv[2]);
    return 1;
  }
  fclose(o);
  printf("st2_c_host: B=%d T=%d -> %zu samples per utterance, workspace %lld bytes\n", B, T, L, (long long)ws_bytes);
  CHECK_ST2(st2_destroy(eng));
  return 0;
}
