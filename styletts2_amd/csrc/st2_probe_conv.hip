// CU health probe: when did every XCD finish, and which compute units ran the conv path's workgroups abnormally slowly
// (st2_probe_cu_health, st2.h) -- with the CU mask that leaves those out.
//
// The probe IS the conv kernel: this translation unit holds a private, instrumented copy of st2_conv1d_xs_impl.h
// (ST2_XS_ABLATE = 64: per-workgroup s_memtime stamps at start / k-loop end / exit + HW_ID / XCC_ID) and runs the launch that
// separated the box classes of rounds 1-3 -- k = 7, C = 256, L = 8 000, B = 32, 128 x 256 tiles, residual + statistics
// epilogue -- on synthetic operands.  History: on the slow class XCD 7 finished that launch at 1 082 us against ~560 for the
// others, with 8 CUs of one shader engine at 10-12 x the epilogue cycles (profiles/r04/r04h1_*, r04j_*); this probe then found the
// same group at 3.5 x on EVERY box, which gave the mechanism away -- those were the row-end tiles of the launch (32 tiles per
// row: tile 31 of every row goes to XCD 7, and to the same CUs) on a slow generic epilogue, not degraded hardware
// (DESIGN.md section 6).  With the epilogue fixed it reads 0 slow CUs; it stays as the check for a genuinely bad CU: CUs
// whose median epilogue takes > 3 x the chip's median are reported and their CU-mask bits found by running an 8-workgroup
// kernel on single-bit-masked streams (the driver's bit -> CU numbering is not documented; measured, not assumed).
// Diagnostic entry point: allocates ~0.9 GB for its duration, synchronises the device.
#define ST2_XS_ABLATE 64
#include "st2_conv1d_xs_impl.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace {

__global__ void pc_fill_planes(_Float16* p, int64_t n, uint32_t seed, float scale) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint32_t h = (uint32_t)i * 2654435761u + seed;
  h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
  p[i] = (_Float16)(((int)(h & 0xffff) - 32768) * (scale / 32768.f));
}
__global__ void pc_fill_f32(float* p, int64_t n, uint32_t seed) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint32_t h = (uint32_t)i * 2654435761u + seed;
  h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
  p[i] = ((int)(h & 0xffff) - 32768) * (1.f / 32768.f);
}
__global__ void pc_whoami(unsigned long long* out) {  // one stamp per workgroup: workgroup b is dispatched to XCD b % 8
  if (threadIdx.x == 0)
    out[blockIdx.x] = (unsigned long long)__builtin_amdgcn_s_getreg(0xF804) |
                      ((unsigned long long)__builtin_amdgcn_s_getreg(0xF814) << 32) | (1ull << 63);
}

// (XCC, SE, SH, CU) of a workgroup from its HW_ID | XCC_ID << 32 stamp
inline unsigned long long cu_key(unsigned long long id) { return (((id >> 32) & 15) << 32) | (id & 0xFF00); }

}  // namespace

extern "C" int st2_probe_cu_health(char* json, int32_t cap, uint32_t* mask_out, int32_t mask_words, int32_t* n_excluded) {
  ST2_REQUIRE(json && cap >= 256 && mask_out && mask_words >= 8 && n_excluded, "st2_probe_cu_health: bad arguments");
  *n_excluded = 0;
  std::vector<void*> bufs;
  auto cleanup = [&]() {
    for (void* p : bufs) (void)hipFree(p);
    bufs.clear();
  };
  auto alloc = [&](size_t bytes) -> void* {
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) {
      (void)hipGetLastError();
      return nullptr;
    }
    bufs.push_back(p);
    return p;
  };
#define HCK(x)                                                                      \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      st2_set_error("st2_probe_cu_health: %s: %s", #x, hipGetErrorString(e_));     \
      cleanup();                                                                    \
      return 1;                                                                     \
    }                                                                               \
  } while (0)
  int dev = 0;
  HCK(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  HCK(hipGetDeviceProperties(&prop, dev));
  const int num_cu = prop.multiProcessorCount;
  const int words = (num_cu + 31) / 32;
  ST2_REQUIRE(words <= mask_words, "st2_probe_cu_health: %d CUs need %d mask words", num_cu, words);
  for (int i = 0; i < mask_words; ++i) mask_out[i] = 0;
  for (int i = 0; i < num_cu; ++i) mask_out[i / 32] |= 1u << (i % 32);

  // ---- the discriminating launch on synthetic operands ----------------------------------------------------------------
  const int ks = 7, C = 256, L = 8000, B = 32, halo = 32;
  const int Lp = halo + (L + 1 + 511) / 512 * 512 + 96, cg = C / 8, pitch = (L + 31) / 32 * 32, nt = (L + 127) / 128;
  const int64_t plane = (int64_t)cg * Lp * 8, wq_halves = (int64_t)(C / 16) * ks * 2 * C * 16, y_elems = (int64_t)B * C * pitch;
  _Float16* xs = (_Float16*)alloc((size_t)B * 2 * plane * 2);
  _Float16* wq = (_Float16*)alloc((size_t)wq_halves * 2);
  float* y = (float*)alloc((size_t)y_elems * 4);
  float* res = (float*)alloc((size_t)y_elems * 4);
  float* vec = (float*)alloc((size_t)C * 2 * 4);
  float* part = (float*)alloc((size_t)B * C * nt * 3 * 4);  // sums + shifts (st2.h: d.part)
  const int64_t n_wg = (int64_t)((L + 255) / 256) * (C / 128) * B;
  unsigned long long* tl = (unsigned long long*)alloc((size_t)n_wg * 64);
  if (!xs || !wq || !y || !res || !vec || !part || !tl) {
    st2_set_error("st2_probe_cu_health: out of device memory");
    cleanup();
    return 1;
  }
  hipLaunchKernelGGL(pc_fill_planes, dim3((unsigned)((B * 2 * plane + 255) / 256)), dim3(256), 0, 0, xs, (int64_t)B * 2 * plane, 17u, 12.f);
  hipLaunchKernelGGL(pc_fill_planes, dim3((unsigned)((wq_halves + 255) / 256)), dim3(256), 0, 0, wq, wq_halves, 5u, 16384.f);
  hipLaunchKernelGGL(pc_fill_f32, dim3((unsigned)((y_elems + 255) / 256)), dim3(256), 0, 0, res, y_elems, 7u);
  hipLaunchKernelGGL(pc_fill_f32, dim3(2), dim3(256), 0, 0, vec, (int64_t)C * 2, 9u);
  HCK(hipMemset(tl, 0, (size_t)n_wg * 64));
  st2_conv_desc d;
  memset(&d, 0, sizeof(d));
  d.B = B; d.C_in = C; d.C_out = C; d.L_in = L; d.L_out = L; d.ks = ks; d.dil = 1; d.pad_left = (ks - 1) / 2;
  d.wq = wq; d.wq_co_pad = C; d.wq_cin_pad = C;
  d.x_scale = 8.f; d.out_scale = 1.f / 8.f; d.w_row_scale = vec + C;
  d.bias = vec;
  d.y = y; d.y_bs = (int64_t)C * pitch; d.y_cs = pitch;
  d.res = res; d.res_bs = (int64_t)C * pitch; d.res_cs = pitch;
  d.div = 1.0f;
  d.xs = xs; d.xs_cg = cg; d.xs_lp = Lp; d.xs_halo = halo;
  d.part = part; d.part_nt = nt;
  d.stats = reinterpret_cast<const float*>(tl);  // ST2_XS_ABLATE & 64: the timeline buffer
  for (int rep = 0; rep < 2; ++rep)  // warm-up + the measured launch (the stamps of the last one stay)
    if (launch<7, 16, 4, 1, 8, 2>(d, 0, false) != 0) {
      cleanup();
      return 1;
    }
  HCK(hipDeviceSynchronize());
  std::vector<unsigned long long> h((size_t)n_wg * 8);
  HCK(hipMemcpy(h.data(), tl, (size_t)n_wg * 64, hipMemcpyDeviceToHost));
  std::map<unsigned long long, std::vector<double>> per_cu;
  std::vector<double> all;
  unsigned long long r_first = ~0ull, r_last = 0;
  double xcd_end[16] = {};
  for (int64_t i = 0; i < n_wg; ++i) {
    if (!h[i * 8 + 1]) continue;
    const double epi = (double)(h[i * 8 + 3] - h[i * 8 + 2]);  // exit - k-loop end, shader cycles
    per_cu[cu_key(h[i * 8])].push_back(epi);
    all.push_back(epi);
    r_first = std::min(r_first, h[i * 8 + 4]);
    r_last = std::max(r_last, h[i * 8 + 5]);
  }
  if (all.empty()) {  // (advisor, round 4) an early return must not leak the probe's ~0.9 GB of buffers
    st2_set_error("st2_probe_cu_health: the probe launch left no stamps");
    cleanup();
    return 1;
  }
  for (int64_t i = 0; i < n_wg; ++i)
    if (h[i * 8 + 1]) {
      const int x = (int)((h[i * 8] >> 32) & 15);
      xcd_end[x] = std::max(xcd_end[x], (double)(h[i * 8 + 5] - r_first) / 100.0);
    }
  std::sort(all.begin(), all.end());
  const double med = all[all.size() / 2];
  struct Slow { unsigned long long key; double x; };
  std::vector<Slow> slow;
  for (auto& kv : per_cu) {
    std::vector<double>& v = kv.second;
    std::sort(v.begin(), v.end());
    const double m = v[v.size() / 2];
    if (m > 3.0 * med) slow.push_back({kv.first, m / med});
  }

  // ---- CU-mask bits of the slow CUs: 8-workgroup kernels on single-bit-masked streams ------------------------------
  std::map<unsigned long long, int> bit_of;  // cu_key | 1 << 63 -> mask bit
  int mapped = 0, map_tried = 0, map_stream_fail = 0, map_run_fail = 0;
  if (!slow.empty()) {
    std::vector<int> xcds;
    for (auto& s : slow) {
      const int x = (int)(s.key >> 32);
      if (std::find(xcds.begin(), xcds.end(), x) == xcds.end()) xcds.push_back(x);
    }
    // Bit i of a CU mask belongs to XCD i % 8 (the driver deals the bits out round-robin over the XCDs); which CU of that XCD
    // it is, is measured: a stream with ONLY that bit set runs an 8-workgroup kernel -- workgroup b goes to XCD b % 8, an XCD
    // whose share of the mask is empty runs unrestricted (observed: r04m1, 288 one-workgroup probes landed on XCD 0 only) --
    // and the workgroup that reports XCD i % 8 names the CU.
    unsigned long long* who = (unsigned long long*)alloc(64);
    if (!who) {
      st2_set_error("st2_probe_cu_health: out of device memory");
      cleanup();
      return 1;
    }
    auto probe_bit = [&](int bit) -> bool {
      std::vector<uint32_t> m(words, 0u);
      m[bit / 32] = 1u << (bit % 32);
      hipStream_t s = nullptr;
      ++map_tried;
      if (hipExtStreamCreateWithCUMask(&s, (uint32_t)words, m.data()) != hipSuccess) {
        (void)hipGetLastError();
        ++map_stream_fail;
        return false;
      }
      unsigned long long v[8] = {};
      bool ok = hipMemsetAsync(who, 0, 64, s) == hipSuccess;
      hipLaunchKernelGGL(pc_whoami, dim3(8), dim3(64), 0, s, who);
      ok = ok && hipGetLastError() == hipSuccess && hipMemcpyAsync(v, who, 64, hipMemcpyDeviceToHost, s) == hipSuccess &&
           hipStreamSynchronize(s) == hipSuccess;
      (void)hipStreamDestroy(s);
      if (!ok) {
        (void)hipGetLastError();
        ++map_run_fail;
        return false;
      }
      bool found = false;
      for (int w = 0; w < 8; ++w)
        if ((v[w] >> 63) && (int)((v[w] >> 32) & 15) == bit % 8) {
          bit_of[cu_key(v[w]) | (1ull << 63)] = bit;
          found = true;
        }
      return found;
    };
    // the driver deals consecutive bits out round-robin over the XCDs: try bits = xcd (mod 8) first, then everything else
    for (int x : xcds)
      for (int bit = x; bit < num_cu; bit += 8) probe_bit(bit);
    bool all_found = true;
    for (auto& s : slow) all_found &= bit_of.count(s.key | (1ull << 63)) != 0;
    if (!all_found)
      for (int bit = 0; bit < num_cu && !all_found; ++bit) {
        probe_bit(bit);
        all_found = true;
        for (auto& s : slow) all_found &= bit_of.count(s.key | (1ull << 63)) != 0;
      }
    // never hand back a mask that excludes more than an eighth of the chip: that is not "a few slow CUs"
    if ((int)slow.size() <= num_cu / 8)
      for (auto& s : slow) {
        auto it = bit_of.find(s.key | (1ull << 63));
        if (it == bit_of.end()) continue;
        mask_out[it->second / 32] &= ~(1u << (it->second % 32));
        ++mapped;
      }
  }
  *n_excluded = mapped;

  std::string js = "{";
  char b[256];
  snprintf(b, sizeof b, "\"launch\": \"st2_conv1d_xs k7 C256 L8000 B32, 128x256 tiles\", \"workgroups\": %lld, \"launch_us\": %.1f, "
           "\"epilogue_cycles_median\": %.0f, \"epilogue_cycles_max\": %.0f, \"xcd_end_us\": [", (long long)all.size(),
           (double)(r_last - r_first) / 100.0, med, all.back());
  js += b;
  for (int x = 0; x < 8; ++x) {
    snprintf(b, sizeof b, "%s%.0f", x ? ", " : "", xcd_end[x]);
    js += b;
  }
  js += "], \"slow_cus\": [";
  for (size_t i = 0; i < slow.size() && i < 64; ++i) {
    auto it = bit_of.find(slow[i].key | (1ull << 63));
    snprintf(b, sizeof b, "%s{\"xcc\": %d, \"se\": %d, \"sh\": %d, \"cu\": %d, \"x_median\": %.1f, \"mask_bit\": %d}", i ? ", " : "",
             (int)(slow[i].key >> 32), (int)((slow[i].key >> 13) & 7), (int)((slow[i].key >> 12) & 1), (int)((slow[i].key >> 8) & 15),
             slow[i].x, it == bit_of.end() ? -1 : it->second);
    js += b;
  }
  snprintf(b, sizeof b, "], \"n_slow_cus\": %d, \"n_excluded\": %d, \"cus\": %d, \"mask_probes\": {\"tried\": %d, "
           "\"stream_create_failed\": %d, \"run_failed\": %d, \"cus_mapped\": %d}}", (int)slow.size(), mapped, num_cu, map_tried,
           map_stream_fail, map_run_fail, (int)bit_of.size());
  js += b;
  cleanup();
  ST2_REQUIRE((int)js.size() + 1 <= cap, "st2_probe_cu_health: output needs %zu bytes", js.size() + 1);
  memcpy(json, js.c_str(), js.size() + 1);
  return 0;
#undef HCK
}
