// Standalone micro-benchmark of the split-f16 MFMA conv kernel (csrc/st2_conv1d_xs_impl.h) on the bench's dominant
// shape classes, without Python / PyTorch: plain hipMalloc buffers, random f16 planes, HIP events around N launches.
// Built once per ablation mask (-DST2_XS_ABLATE=m, see the header) by tools/build_xs_bench.sh:
//   ./xs_bench_<m> [ks=11] [dil=1] [C=128] [L=48001] [B=32] [res=1] [stats=1] [reps=10] [zero_data=0]
#include "../styletts2_amd/csrc/st2_conv1d_xs_impl.h"

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

int main(int argc, char** argv) {
  auto arg = [&](int i, int def) { return argc > i ? atoi(argv[i]) : def; };
  const int ks = arg(1, 11), dil = arg(2, 1), C = arg(3, 128), L = arg(4, 48001), B = arg(5, 32);
  const int use_res = arg(6, 1), use_stats = arg(7, 1), reps = arg(8, 10);
  const int xs_variant = getenv("XS_VARIANT") ? atoi(getenv("XS_VARIANT")) : -1;  // st2xs::XS_V_* bits, -1 = the rule
  const float dscale = arg(9, 0) ? 0.f : 1.f;  // 10th argument 1: all-zero operands (what does the data cost in clock?)
  const int chunk = ks <= 3 ? 32 : 16;
  const int C_pad = (C + chunk - 1) / chunk * chunk;
  const int co_blk = C > 64 ? 128 : (C > 32 ? 64 : 32);
  const int co_pad = (C + co_blk - 1) / co_blk * co_blk;
  const int halo = 32;
  const int Lp = halo + (L + 1 + 511) / 512 * 512 + 96;
  const int cg = (C + 31) / 32 * 32 / 8;
  const int pitch = (L + 31) / 32 * 32;
  const int64_t xs_halves = (int64_t)B * 2 * cg * Lp * 8;
  const int64_t wq_halves = (int64_t)(C_pad / 16) * ks * 2 * co_pad * 16;
  const int64_t y_elems = (int64_t)B * C * pitch;
  _Float16 *xs, *wq;
  float *y, *res, *bias, *rsc, *part;
  CK(hipMalloc(&xs, xs_halves * 2));
  CK(hipMalloc(&wq, wq_halves * 2));
  CK(hipMalloc(&y, y_elems * 4));
  CK(hipMalloc(&res, y_elems * 4));
  CK(hipMalloc(&bias, co_pad * 4));
  CK(hipMalloc(&rsc, co_pad * 4));
  int nt = (L + 31) / 32;  // allocated for the narrowest slot width; set per build below
  CK(hipMalloc(&part, (int64_t)B * C * nt * 3 * 4));  // (sum, sumsq) per slot, then the slots' shifts
  CK(hipMemset(part, 0, (int64_t)B * C * nt * 3 * 4));
  // hi plane ~ values in +-24 (x8 scaled activations), lo plane ~ 2^-11 of that; weights hi in +-16384, lo in +-8
  const int64_t plane = (int64_t)cg * Lp * 8;
  for (int b = 0; b < B; ++b) {
    hipLaunchKernelGGL(fill_planes, dim3((plane + 255) / 256), dim3(256), 0, 0, xs + (int64_t)b * 2 * plane, plane, 17u + b, 24.f * dscale);
    hipLaunchKernelGGL(fill_planes, dim3((plane + 255) / 256), dim3(256), 0, 0, xs + (int64_t)b * 2 * plane + plane, plane, 91u + b, 0.012f * dscale);
  }
  hipLaunchKernelGGL(fill_planes, dim3((wq_halves + 255) / 256), dim3(256), 0, 0, wq, wq_halves, 5u, 16384.f * dscale);
  hipLaunchKernelGGL(fill_f32, dim3((y_elems + 255) / 256), dim3(256), 0, 0, res, y_elems, 7u);
  hipLaunchKernelGGL(fill_f32, dim3(1), dim3(256), 0, 0, bias, (int64_t)co_pad, 9u);
  hipLaunchKernelGGL(fill_f32, dim3(1), dim3(256), 0, 0, rsc, (int64_t)co_pad, 11u);
  CK(hipDeviceSynchronize());

  st2_conv_desc d;
  memset(&d, 0, sizeof(d));
  d.B = B; d.C_in = C; d.C_out = C; d.L_in = L; d.L_out = L; d.ks = ks; d.dil = dil; d.pad_left = (ks - 1) * dil / 2;
  d.wq = wq; d.wq_co_pad = co_pad; d.wq_cin_pad = C_pad;
  d.x_scale = 8.f; d.out_scale = 1.f / 8.f; d.w_row_scale = rsc;
  d.bias = bias;
  d.y = y; d.y_bs = (int64_t)C * pitch; d.y_cs = pitch;
  if (use_res) { d.res = res; d.res_bs = (int64_t)C * pitch; d.res_cs = pitch; }
  d.div = 1.0f;
  d.xs = xs; d.xs_cg = cg; d.xs_lp = Lp; d.xs_halo = halo;
  {  // slot width of the partial sums: forced small-grid builds (XS_VARIANT bits 2 / 3) or the library's geometry rule
    int pc = (xs_variant >= 0 && (xs_variant & 8)) ? 32 : ((xs_variant >= 0 && (xs_variant & 4)) ? 64 : 128);
    if (xs_variant < 0) pc = st2xs::small_grid_cols(d);
    nt = (L + pc - 1) / pc;
    if (use_stats) { d.part = part; d.part_nt = nt; d.part_cols = pc; }
    printf("tile columns %d (%d workgroups)\n", pc < 128 ? pc : 128, ((L + (pc < 128 ? pc : 128) - 1) / (pc < 128 ? pc : 128)) * ((C + 127) / 128) * B);
  }

  unsigned long long* tl = nullptr;
  const int64_t n_wg = (int64_t)((L + 127) / 128) * ((C + 127) / 128) * B * 4;  // upper bound on workgroups
  if (ST2_XS_ABLATE & 64) {
    CK(hipMalloc(&tl, n_wg * 64));
    CK(hipMemset(tl, 0, n_wg * 64));
    d.stats = reinterpret_cast<const float*>(tl);
  }
  auto run = [&]() -> int {
#ifdef XS_BENCH_KS  // build one kernel size only (compile time)
    if (ks != XS_BENCH_KS) { fprintf(stderr, "this binary was built for ks = %d\n", XS_BENCH_KS); return 1; }
    return st2xs::launch_by_cout<XS_BENCH_KS, (XS_BENCH_KS <= 3 ? 32 : 16)>(d, 0, xs_variant);
#else
    switch (ks) {
      case 3: return st2xs::launch_by_cout<3, 32>(d, 0, xs_variant);
      case 7: return st2xs::launch_by_cout<7, 16>(d, 0, xs_variant);
      case 11: return st2xs::launch_by_cout<11, 16>(d, 0, xs_variant);
      default: fprintf(stderr, "ks must be 3, 7 or 11\n"); return 1;
    }
#endif
  };
  for (int i = 0; i < 2; ++i)
    if (run()) return 1;
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i)
    if (run()) return 1;
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  const double flop = 2.0 * B * C * (double)C * ks * L;
  printf("xs_bench abl=%d ks=%d dil=%d C=%d L=%d B=%d res=%d stats=%d: %.4f ms / launch, %.1f algorithmic TFLOP/s "
         "(%.3f of 833)\n", ST2_XS_ABLATE, ks, dil, C, L, B, use_res, use_stats, ms, flop / ms / 1e9,
         flop / ms / 1e9 / (2500.0 / 3));
  {  // FNV-1a over the bits of the valid part of y and of the partial sums: builds that claim bitwise equality print the same
    std::vector<float> hy((size_t)B * C * pitch), hp((size_t)B * C * nt * 3);  // sums, then shifts
    CK(hipMemcpy(hy.data(), y, hy.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hp.data(), part, hp.size() * 4, hipMemcpyDeviceToHost));
    unsigned long long hsh = 1469598103934665603ull;
    auto mix = [&](float v) { unsigned u; memcpy(&u, &v, 4); hsh = (hsh ^ u) * 1099511628211ull; };
    for (int64_t r = 0; r < (int64_t)B * C; ++r)
      for (int l = 0; l < L; ++l) mix(hy[(size_t)r * pitch + l]);
    printf("checksum_y %016llx\n", hsh);
    if (use_stats) {
      double s1 = 0, s2 = 0;
      for (size_t i = 0; i + 1 < (size_t)B * C * nt * 2; i += 2) { s1 += hp[i]; s2 += hp[i + 1]; }
      for (float v : hp) mix(v);
      printf("partial sums: total %.9g / %.9g over %d slots per row\n", s1, s2, nt);
    }
    printf("checksum %016llx\n", hsh);
  }
  if (ST2_XS_ABLATE & 64) {  // dump the last launch's timeline: one line per workgroup
    std::vector<unsigned long long> h(n_wg * 8);
    CK(hipMemcpy(h.data(), tl, n_wg * 64, hipMemcpyDeviceToHost));
    const char* path = argc > 10 ? argv[10] : "xs_timeline.txt";
    FILE* f = fopen(path, "w");
    if (!f) return 1;
    for (int64_t i = 0; i < n_wg; ++i)
      if (h[i * 8 + 1]) fprintf(f, "%lld %llx %llu %llu %llu %llu %llu\n", (long long)i, h[i * 8], h[i * 8 + 1], h[i * 8 + 2], h[i * 8 + 3], h[i * 8 + 4], h[i * 8 + 5]);
    fclose(f);
    printf("timeline written to %s\n", path);
  }
  return 0;
}
