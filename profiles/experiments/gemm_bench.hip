// (k3_bench: the same harness on the k = 3 resblock / decoder-front convs -- built from this file with -DK3_BENCH)
// Micro-benchmark of the k = 1 ("token GEMM") launches of the split-f16 conv kernel: the denoiser's / PL-BERT's Linears run
// as Conv1d(k = 1) over the B*N merged tokens (M = C_out in 512..2304, K = C_in in 512..2048, N = 3200 tokens at B = 32) and
// are the third-largest kernel of a bench step (200 launches, 9.5 ms at 0.17-0.28 of the roof: 200 workgroups of 128 x 128
// on 256 CUs, one per CU, nothing to hide the staging latency behind).  Times the library's kernel template at other tile
// shapes / chunk depths on those shapes, no PyTorch:
//   ./gemm_bench [C_out=2048] [C_in=1024] [L=3200] [B=1] [reps=20]
#include "../../styletts2_amd/csrc/st2_conv1d_xs_impl.h"
#define ST2_STATUS_GEMM_TIMEOUT 8  // experiment-only status bit
#include "gemm_sk_experiment.h"  // persistent stream-K build: measured, not adopted (profiles/LAB_NOTES.md, round 4)

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

void st2_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vfprintf(stderr, fmt, ap);
  va_end(ap);
  fputc('\n', stderr);
}
int* st2_status_device_ptr() { return nullptr; }

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                  \
      return 1;                                                                \
    }                                                                          \
  } while (0)

__global__ void fill_planes(_Float16* p, int64_t n, uint32_t seed, float scale) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint32_t h = (uint32_t)i * 2654435761u + seed;
  h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
  p[i] = (_Float16)(((int)(h & 0xffff) - 32768) * (scale / 32768.f));
}
__global__ void fill_f32(float* p, int64_t n, uint32_t seed) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint32_t h = (uint32_t)i * 2654435761u + seed;
  h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
  p[i] = ((int)(h & 0xffff) - 32768) * (1.f / 32768.f);
}

template <int KS, int CI_T, int WM, int WN, int TN, int OCC>
int lx(const st2_conv_desc& d, hipStream_t s) {
  return launch<KS, CI_T, WM, WN, TN, OCC>(d, s, false);
}
static int g_num_cu = 256;
template <int CI_T>
int launch_sk(const st2_conv_desc& d, hipStream_t s) {
  const int w = st2sk::pick_workers(d, g_num_cu, CI_T);
  if (!w) return 1;
  return st2sk::launch<CI_T>(d, w, s);
}

struct Variant {
  const char* name;
  int (*fn)(const st2_conv_desc&, hipStream_t);
  int co_blk, ci_t;
};

int main(int argc, char** argv) {
  auto arg = [&](int i, int def) { return argc > i ? atoi(argv[i]) : def; };
  const int Co = arg(1, 2048), Ci = arg(2, 1024), L = arg(3, 3200), B = arg(4, 1), reps = arg(5, 20);
#ifdef K3_BENCH
  const Variant vars[] = {
      {"128x128 c32 occ3 (library)", &lx<3, 32, 4, 1, 4, 3>, 128, 32},
      {"128x128 c32 occ2", &lx<3, 32, 4, 1, 4, 2>, 128, 32},
      {"128x64  c32 occ3", &lx<3, 32, 4, 1, 2, 3>, 128, 32},
      {"128x64  c32 occ4", &lx<3, 32, 4, 1, 2, 4>, 128, 32},
      {"64x128  c32 occ3", &lx<3, 32, 2, 2, 2, 3>, 64, 32},
      {"64x128  c32 occ4", &lx<3, 32, 2, 2, 2, 4>, 64, 32},
      {"64x64   c32 occ4", &lx<3, 32, 2, 2, 1, 4>, 64, 32},
      {"128x256 c32 occ2", &lx<3, 32, 4, 1, 8, 2>, 128, 32},
      {"128x128 c16 occ3", &lx<3, 16, 4, 1, 4, 3>, 128, 16},
      {"128x64  c16 occ4", &lx<3, 16, 4, 1, 2, 4>, 128, 16},
  };
  const int KSZ = 3;
#else
  const int KSZ = 1;
  const Variant vars[] = {
      {"128x128 c32 occ3 (library)", &lx<1, 32, 4, 1, 4, 3>, 128, 32},
      {"128x128 c32 occ2", &lx<1, 32, 4, 1, 4, 2>, 128, 32},
      {"128x128 c64 occ3", &lx<1, 64, 4, 1, 4, 3>, 128, 64},
      {"128x64  c32 occ3", &lx<1, 32, 4, 1, 2, 3>, 128, 32},
      {"128x64  c64 occ3", &lx<1, 64, 4, 1, 2, 3>, 128, 64},
      {"64x128  c32 occ3", &lx<1, 32, 2, 2, 2, 3>, 64, 32},
      {"64x128  c64 occ3", &lx<1, 64, 2, 2, 2, 3>, 64, 64},
      {"64x64   c32 occ3", &lx<1, 32, 2, 2, 1, 3>, 64, 32},
      {"64x64   c64 occ3", &lx<1, 64, 2, 2, 1, 3>, 64, 64},
      {"128x32  c64 occ3", &lx<1, 64, 4, 1, 1, 3>, 128, 64},
      {"stream-K 256x128 c32", &launch_sk<32>, 128, 32},
      {"stream-K 256x128 c64", &launch_sk<64>, 128, 64},
  };
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) == hipSuccess) g_num_cu = prop.multiProcessorCount;
    if (getenv("SK_WORKERS")) g_num_cu = atoi(getenv("SK_WORKERS"));
  }
#endif
  const int halo = 32;
  const int Lp = halo + (L + 1 + 511) / 512 * 512 + 96;
  const int C_pad = (Ci + 63) / 64 * 64;
  const int cg = C_pad / 8;
  const int co_pad = (Co + 127) / 128 * 128;
  const int pitch = (L + 31) / 32 * 32;
  const int64_t plane = (int64_t)cg * Lp * 8;
  const int64_t wq_halves = (int64_t)(C_pad / 16) * KSZ * 2 * co_pad * 16;
  const int64_t y_elems = (int64_t)B * Co * pitch;
  _Float16 *xs, *wq;
  float *y, *bias, *rsc;
  CK(hipMalloc(&xs, (int64_t)B * 2 * plane * 2));
  CK(hipMalloc(&wq, wq_halves * 2));
  CK(hipMalloc(&y, y_elems * 4));
  CK(hipMalloc(&bias, co_pad * 4));
  CK(hipMalloc(&rsc, co_pad * 4));
  for (int b = 0; b < B; ++b) {
    hipLaunchKernelGGL(fill_planes, dim3((plane + 255) / 256), dim3(256), 0, 0, xs + (int64_t)b * 2 * plane, plane, 17u + b, 24.f);
    hipLaunchKernelGGL(fill_planes, dim3((plane + 255) / 256), dim3(256), 0, 0, xs + (int64_t)b * 2 * plane + plane, plane, 91u + b, 0.012f);
  }
  hipLaunchKernelGGL(fill_planes, dim3((wq_halves + 255) / 256), dim3(256), 0, 0, wq, wq_halves, 5u, 16384.f);
  hipLaunchKernelGGL(fill_f32, dim3((co_pad + 255) / 256), dim3(256), 0, 0, bias, (int64_t)co_pad, 9u);
  hipLaunchKernelGGL(fill_f32, dim3((co_pad + 255) / 256), dim3(256), 0, 0, rsc, (int64_t)co_pad, 11u);
  CK(hipDeviceSynchronize());
  const double flop = 2.0 * B * Co * (double)Ci * L * KSZ;
  const int use_res = arg(6, 0), use_part = arg(7, 0);
  float *res = nullptr, *part = nullptr;
  if (use_res) { CK(hipMalloc(&res, y_elems * 4)); hipLaunchKernelGGL(fill_f32, dim3((y_elems + 255) / 256), dim3(256), 0, 0, res, y_elems, 7u); }
  if (use_part) CK(hipMalloc(&part, (int64_t)B * Co * ((L + 127) / 128) * 2 * 4));
  std::vector<float> ref;
  void* sk_ws = nullptr;
  const int64_t sk_bytes = st2sk::workspace_bytes(st2sk::MAX_WORKERS);
  CK(hipMalloc(&sk_ws, sk_bytes));
  CK(hipMemset(sk_ws, 0, st2sk::FLAG_BYTES));
  for (const Variant& v : vars) {
    st2_conv_desc d;
    memset(&d, 0, sizeof(d));
    d.B = B; d.C_in = Ci; d.C_out = Co; d.L_in = L; d.L_out = L; d.ks = KSZ; d.dil = 1; d.pad_left = (KSZ - 1) / 2;
    d.wq = wq; d.wq_co_pad = co_pad; d.wq_cin_pad = (Ci + v.ci_t - 1) / v.ci_t * v.ci_t;
    d.x_scale = 8.f; d.out_scale = 1.f / 8.f; d.w_row_scale = rsc;
    d.bias = bias;
    d.y = y; d.y_bs = (int64_t)Co * pitch; d.y_cs = pitch;
    d.div = 1.0f;
    d.xs = xs; d.xs_cg = cg; d.xs_lp = Lp; d.xs_halo = halo;
    d.splitk_ws = sk_ws; d.splitk_ws_bytes = sk_bytes;
    if (use_res) { d.res = res; d.res_bs = (int64_t)Co * pitch; d.res_cs = pitch; }
    if (use_part && v.co_blk >= 0) { d.part = part; d.part_nt = (L + 127) / 128; }
    CK(hipMemset(y, 0, y_elems * 4));
    bool bad = false;
    for (int i = 0; i < 2 && !bad; ++i) bad = v.fn(d, 0) != 0;
    if (bad) { printf("%-28s refused\n", v.name); continue; }
    CK(hipDeviceSynchronize());
    std::vector<float> h((size_t)Co * pitch);  // first batch item
    CK(hipMemcpy(h.data(), y, h.size() * 4, hipMemcpyDeviceToHost));
    double md = 0;
    if (ref.empty()) ref = h;
    else {
      double mr = 0;
      for (int co = 0; co < Co; ++co) for (int l = 0; l < L; ++l) {
        md = fmax(md, fabs((double)h[(size_t)co * pitch + l] - ref[(size_t)co * pitch + l]));
        mr = fmax(mr, fabs((double)ref[(size_t)co * pitch + l]));
      }
      md /= fmax(mr, 1e-30);  // relative to the largest output
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) v.fn(d, 0);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    printf("%s M=%d K=%d N=%d B=%d res=%d part=%d  %-28s %8.1f us  %6.1f TFLOP/s (%.3f of 833)  max|dy| / max|y| vs library %.2e\n", KSZ == 3 ? "k3_bench" : "gemm_bench", Co, Ci, L, B, use_res, use_part,
           v.name, ms * 1e3, flop / ms / 1e9, flop / ms / 1e9 / (2500.0 / 3), md);
  }
  return 0;
}
