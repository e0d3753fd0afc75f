// Stand-alone reproducer (no Python, no torch) for the round-5 observation "a BiLSTM on one queue returns different bits while
// narrow-tile xs convs run on another queue" (profiles/LAB_NOTES.md; GPUTEST_r05: 3 / 30 cooperative calls next to the k = 3
// 32-column build).  Two kinds of victim on stream A, a matrix of aggressors on stream B:
//
//   victims   synth_single  the single-CU recurrence's access pattern (csrc/st2_lstm.hip: thread j sweeps W[k][g*H + j], k = 0..255,
//                           4 gates, unrolled by 8) over a 2 MB READ-ONLY buffer whose every word is a function of its index, so
//                           each loaded word is checked in registers -- no second memory access, no reference run -- and a wrong
//                           word is LOGGED with (call, step, workgroup, thread, k, gate, value, an immediate plain re-read, an
//                           agent-scope re-read, HW_ID, XCC_ID, microseconds since kernel start)
//             synth_coop    the cooperative kernel's one-off slice load (csrc/st2_lstm_coop.hip: 128 words per thread at kernel
//                           start), preceded by the scratch-clearing kernel as in the product; variants _sc1 (agent-scope loads:
//                           bypass the CU's vector L1) and _late (the loads start 20 us into the kernel)
//             real_single / real_coop   the product kernels themselves through the C ABI of styletts2_amd/libst2_hip.so (dlopen):
//                           st2_lstm_bidir / st2_lstm_bidir_coop_recovering on fixed inputs, outputs compared bitwise with an
//                           idle reference run
//   aggressors  none | xs_k{3,7,11}_n{32,64,128} (this tree's conv kernel, forced tile width, with partial sums as in the canary)
//               | empty (kernel boundaries only) | samead (every lane re-reads one 16-byte slot: what 50-70 % of a narrow tile's
//               staging lanes do) | mfma (matrix-pipe bursts, no memory) | stream (small-grid streaming read)
//
// Built in three flavours by tools/build_lstm_repro.sh: default, -DST2_XS_SETPRIO=0, -DST2_XS_PRED_STAGE=1 (bisecting the conv
// kernel's own features).  REPRO_MASK=split puts the two streams on disjoint halves of the CU mask (hipExtStreamCreateWithCUMask).
//   ./lstm_load_repro [victims=all|comma list] [aggressors=all|comma list] [calls=200] [trials=3] [B=1] [N=24]
#include "../styletts2_amd/csrc/st2_conv1d_xs_impl.h"

#include <dlfcn.h>
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

void st2_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vfprintf(stderr, fmt, ap);
  va_end(ap);
  fputc('\n', stderr);
}
int* st2_status_device_ptr() { return nullptr; }

#define CK(x)                                                                                  \
  do {                                                                                         \
    hipError_t e_ = (x);                                                                       \
    if (e_ != hipSuccess) {                                                                    \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));        \
      exit(1);                                                                                 \
    }                                                                                          \
  } while (0)

constexpr int LH = 256;
constexpr uint32_t WMUL = 2654435761u, WADD = 0x9e3779b9u;
__host__ __device__ inline uint32_t wval(uint32_t i) { return i * WMUL + WADD; }

struct LogRec {
  uint32_t call, step, wg, tid, k, gate, got, want, reread, reread_sc1, hwid, xcc;
  uint32_t us_since_start, pad;
};

__device__ __forceinline__ void log_bad(LogRec* log, int* nlog, int maxlog, uint32_t call, uint32_t step, uint32_t wg, uint32_t tid,
                                        uint32_t k, uint32_t gate, uint32_t got, const uint32_t* addr, uint32_t idx,
                                        unsigned long long t0) {
  const int slot = atomicAdd(nlog, 1);
  if (slot >= maxlog) return;
  LogRec r;
  r.call = call; r.step = step; r.wg = wg; r.tid = tid; r.k = k; r.gate = gate; r.got = got; r.want = wval(idx);
  r.reread = *reinterpret_cast<const volatile uint32_t*>(addr);
  r.reread_sc1 = __hip_atomic_load(addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  r.hwid = __builtin_amdgcn_s_getreg(0xF804);
  r.xcc = __builtin_amdgcn_s_getreg(0xF814);
  r.us_since_start = (uint32_t)((__builtin_amdgcn_s_memrealtime() - t0) / 100);  // 100 MHz counter
  r.pad = 0;
  log[slot] = r;
}

// ---- victim 1: the single-CU recurrence's W_hh sweep ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void victim_single(const uint32_t* __restrict__ W, int steps, int call, LogRec* log, int* nlog,
                                                     int maxlog, uint32_t* sink) {
  const int j = threadIdx.x;
  const int dir = blockIdx.y;
  const uint32_t* Wd = W + (size_t)dir * LH * 4 * LH;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  uint32_t acc = 0;
  for (int s = 0; s < steps; ++s) {
    for (int k0 = 0; k0 < LH; k0 += 8) {
      uint32_t v[8][4];
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        const uint32_t* wk = Wd + (size_t)(k0 + kk) * 4 * LH + j;
#pragma unroll
        for (int g = 0; g < 4; ++g) v[kk][g] = wk[g * LH];
      }
      uint32_t bad = 0;
#pragma unroll
      for (int kk = 0; kk < 8; ++kk)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const uint32_t idx = (uint32_t)(dir * LH * 4 * LH + (k0 + kk) * 4 * LH + g * LH + j);
          bad |= (v[kk][g] != wval(idx)) ? (1u << (kk * 4 + g)) : 0u;
          acc += v[kk][g];
        }
      if (bad) {
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
#pragma unroll
          for (int g = 0; g < 4; ++g)
            if (bad & (1u << (kk * 4 + g))) {
              const uint32_t idx = (uint32_t)(dir * LH * 4 * LH + (k0 + kk) * 4 * LH + g * LH + j);
              log_bad(log, nlog, maxlog, call, s, blockIdx.x + gridDim.x * blockIdx.y, j, k0 + kk, g, v[kk][g], W + idx, idx, t0);
            }
      }
    }
    asm volatile("" ::: "memory");
  }
  if (acc == 0x1234567u) sink[0] = acc;
}

// ---- victim 2: the cooperative kernel's register-resident slice, loaded once at kernel start ------------------------------------
// LOADK: 0 plain loads (the product), 1 agent-scope atomic loads (sc1: served by L2, not by the CU's vector L1)
template <int LOADK>
__global__ __launch_bounds__(256) void victim_coop(const uint32_t* __restrict__ W, int call, int delay_us, int hold_us, LogRec* log,
                                                   int* nlog, int maxlog, uint32_t* sink) {
  const int tid = threadIdx.x;
  const int sl = blockIdx.x, dir = blockIdx.z;
  const int unit = tid & 31, kq = tid >> 5;
  const int hu = sl * 32 + unit;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if (delay_us > 0)
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)delay_us * 100) __builtin_amdgcn_s_sleep(8);
  const uint32_t* Wd = W + (size_t)dir * LH * 4 * LH;
  uint32_t w[32][4];
#pragma unroll
  for (int kk = 0; kk < 32; ++kk) {
    const uint32_t* wr = Wd + (size_t)(kq * 32 + kk) * 4 * LH + hu;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if constexpr (LOADK == 1)
        w[kk][g] = __hip_atomic_load(wr + g * LH, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else
        w[kk][g] = wr[g * LH];
    }
  }
  uint32_t acc = 0;
#pragma unroll
  for (int kk = 0; kk < 32; ++kk)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const uint32_t idx = (uint32_t)(dir * LH * 4 * LH + (kq * 32 + kk) * 4 * LH + g * LH + hu);
      acc += w[kk][g];
      if (w[kk][g] != wval(idx))
        log_bad(log, nlog, maxlog, call, 0, sl + 8 * (blockIdx.y + gridDim.y * dir), tid, kq * 32 + kk, g, w[kk][g], W + idx, idx, t0);
    }
  if (hold_us > 0)
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)(delay_us + hold_us) * 100) __builtin_amdgcn_s_sleep(8);
  if (acc == 0x1234567u) sink[0] = acc;
}

__global__ __launch_bounds__(256) void zero16_kernel(uint4* p, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = make_uint4(0u, 0u, 0u, 0u);
}

// ---- synthetic aggressors ----------------------------------------------------------------------------------------------------------
__global__ void aggr_empty() {}

__global__ __launch_bounds__(256) void aggr_samead(const uint4* __restrict__ p, uint4* out, int iters, int stride) {
  uint4 a = make_uint4(0, 0, 0, 0);
  const uint4* q = p + (size_t)blockIdx.x * stride;  // one address per workgroup and iteration, all 256 lanes
  for (int i = 0; i < iters; ++i) {
    const uint4 v = q[(size_t)i * 8];
    a.x ^= v.x; a.y ^= v.y; a.z ^= v.z; a.w ^= v.w;
    asm volatile("" ::: "memory");
  }
  if (a.x == 0x1234567u) out[0] = a;
}

__global__ __launch_bounds__(256) void aggr_mfma(float* out, int iters) {
  f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  h8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = (_Float16)(0.001f * (threadIdx.x + i));
    b[i] = (_Float16)(0.002f * (threadIdx.x ^ i));
  }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
  }
  float t = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) t += acc[j][r];
  if (t == 12345.678f) out[threadIdx.x] = t;
}

__global__ __launch_bounds__(256) void aggr_stream(const uint4* __restrict__ p, size_t n, uint4* out) {
  uint4 a = make_uint4(0, 0, 0, 0);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const uint4 v = p[i];
    a.x ^= v.x; a.y ^= v.y; a.z ^= v.z; a.w ^= v.w;
  }
  if (a.x == 0x1234567u) out[0] = a;
}

__global__ void fill_planes(_Float16* p, int64_t n, uint32_t seed, float scale) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint32_t h = (uint32_t)i * 2654435761u + seed;
  h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
  p[i] = (_Float16)(((int)(h & 0xffff) - 32768) * (scale / 32768.f));
}
__global__ void fill_f32(float* p, int64_t n, uint32_t seed, float scale) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint32_t h = (uint32_t)i * 2654435761u + seed;
  h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
  p[i] = ((int)(h & 0xffff) - 32768) * (scale / 32768.f);
}

// ---- the conv aggressor: one descriptor per (ks, tile width), shape of the canary (B = 1, C = 256, L = 5 680) -------------------
struct ConvLoad {
  st2_conv_desc d;
  int ks, cols;
};

static ConvLoad make_conv(int ks, int cols, int C, int L, int B) {
  ConvLoad c;
  memset(&c, 0, sizeof(c));
  c.ks = ks;
  c.cols = cols;
  const int chunk = ks <= 3 ? 32 : 16;
  const int C_pad = (C + chunk - 1) / chunk * chunk;
  const int co_pad = (C + 127) / 128 * 128;
  const int halo = 32;
  const int Lp = halo + (L + 1 + 511) / 512 * 512 + 96;
  const int cg = (C + 31) / 32 * 32 / 8;
  const int pitch = (L + 31) / 32 * 32;
  const int64_t plane = (int64_t)cg * Lp * 8;
  const int64_t xs_halves = (int64_t)B * 2 * plane;
  const int64_t wq_halves = (int64_t)(C_pad / 16) * ks * 2 * co_pad * 16;
  const int64_t y_elems = (int64_t)B * C * pitch;
  _Float16 *xs, *wq;
  float *y, *bias, *rsc, *part;
  CK(hipMalloc(&xs, xs_halves * 2));
  CK(hipMalloc(&wq, wq_halves * 2));
  CK(hipMalloc(&y, y_elems * 4));
  CK(hipMalloc(&bias, co_pad * 4));
  CK(hipMalloc(&rsc, co_pad * 4));
  const int nt = (L + cols - 1) / cols;
  CK(hipMalloc(&part, (int64_t)B * C * nt * 3 * 4));
  CK(hipMemset(part, 0, (int64_t)B * C * nt * 3 * 4));
  for (int b = 0; b < B; ++b) {
    hipLaunchKernelGGL(fill_planes, dim3((plane + 255) / 256), dim3(256), 0, 0, xs + (int64_t)b * 2 * plane, plane, 17u + b, 24.f);
    hipLaunchKernelGGL(fill_planes, dim3((plane + 255) / 256), dim3(256), 0, 0, xs + (int64_t)b * 2 * plane + plane, plane, 91u + b, 0.012f);
  }
  hipLaunchKernelGGL(fill_planes, dim3((wq_halves + 255) / 256), dim3(256), 0, 0, wq, wq_halves, 5u, 16384.f);
  hipLaunchKernelGGL(fill_f32, dim3(1), dim3(256), 0, 0, bias, (int64_t)co_pad, 9u, 1.f);
  hipLaunchKernelGGL(fill_f32, dim3(1), dim3(256), 0, 0, rsc, (int64_t)co_pad, 11u, 1.f);
  CK(hipDeviceSynchronize());
  st2_conv_desc& d = c.d;
  d.B = B; d.C_in = C; d.C_out = C; d.L_in = L; d.L_out = L; d.ks = ks; d.dil = 1; d.pad_left = (ks - 1) / 2;
  d.wq = wq; d.wq_co_pad = co_pad; d.wq_cin_pad = C_pad;
  d.x_scale = 8.f; d.out_scale = 1.f / 8.f; d.w_row_scale = rsc;
  d.bias = bias;
  d.y = y; d.y_bs = (int64_t)C * pitch; d.y_cs = pitch;
  d.div = 1.0f;
  d.xs = xs; d.xs_cg = cg; d.xs_lp = Lp; d.xs_halo = halo;
  d.part = part; d.part_nt = nt; d.part_cols = cols;
  return c;
}

static int launch_conv(const ConvLoad& c, hipStream_t s) {
  const int v = c.cols == 32 ? st2xs::XS_V_N32 : (c.cols == 64 ? st2xs::XS_V_N64 : 0);
  switch (c.ks) {
    case 3: return st2xs::launch_by_cout<3, 32>(c.d, s, v);
    case 7: return st2xs::launch_by_cout<7, 16>(c.d, s, v);
    default: return st2xs::launch_by_cout<11, 16>(c.d, s, v);
  }
}

// ---- the product's BiLSTM kernels through the C ABI -----------------------------------------------------------------------------
typedef int (*lstm_single_fn)(const float*, int64_t, int32_t, const float*, const int32_t*, int32_t, int32_t, int32_t, float*, int64_t,
                              int32_t, void*);
typedef int (*lstm_coop_fn)(const float*, int64_t, int32_t, const float*, const int32_t*, int32_t, int32_t, int32_t, float*, int64_t,
                            int32_t, void*, int64_t, void*);
typedef int64_t (*scratch_fn)(int32_t);
typedef int (*status_fn)(int);

static std::vector<std::string> split(const char* s) {
  std::vector<std::string> out;
  std::string cur;
  for (const char* p = s; *p; ++p) {
    if (*p == ',') { out.push_back(cur); cur.clear(); } else cur.push_back(*p);
  }
  if (!cur.empty()) out.push_back(cur);
  return out;
}

int main(int argc, char** argv) {
  const std::vector<std::string> all_v = {"synth_single", "synth_coop", "synth_coop_sc1", "synth_coop_late", "real_single", "real_coop"};
  const std::vector<std::string> all_a = {"none", "xs_k3_n32", "xs_k3_n64", "xs_k3_n128", "xs_k7_n32", "xs_k7_n128", "xs_k11_n32",
                                          "empty", "samead", "mfma", "stream"};
  std::vector<std::string> victims = (argc > 1 && strcmp(argv[1], "all")) ? split(argv[1]) : all_v;
  std::vector<std::string> aggrs = (argc > 2 && strcmp(argv[2], "all")) ? split(argv[2]) : all_a;
  const int calls = argc > 3 ? atoi(argv[3]) : 200;
  const int trials = argc > 4 ? atoi(argv[4]) : 3;
  const int B = argc > 5 ? atoi(argv[5]) : 1;
  const int N = argc > 6 ? atoi(argv[6]) : 24;
  const char* mask_mode = getenv("REPRO_MASK");
  printf("lstm_load_repro: SETPRIO=%d PRED_STAGE=%d mask=%s calls=%d trials=%d B=%d N=%d\n", ST2_XS_SETPRIO, ST2_XS_PRED_STAGE,
         mask_mode ? mask_mode : "none", calls, trials, B, N);

  hipStream_t sa, sb;
  if (mask_mode && !strcmp(mask_mode, "split")) {
    uint32_t lo[8] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0, 0, 0, 0};
    uint32_t hi[8] = {0, 0, 0, 0, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
    CK(hipExtStreamCreateWithCUMask(&sa, 8, lo));
    CK(hipExtStreamCreateWithCUMask(&sb, 8, hi));
  } else {
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  }

  // the read-only victim buffer: host -> device copy, as torch's .to(device) does
  const size_t WN = (size_t)2 * LH * 4 * LH;
  std::vector<uint32_t> hW(WN);
  for (size_t i = 0; i < WN; ++i) hW[i] = wval((uint32_t)i);
  uint32_t* W;
  CK(hipMalloc(&W, WN * 4));
  CK(hipMemcpy(W, hW.data(), WN * 4, hipMemcpyHostToDevice));
  LogRec* log;
  int* nlog;
  uint32_t* sink;
  const int maxlog = 4096;
  CK(hipMalloc(&log, sizeof(LogRec) * maxlog));
  CK(hipMalloc(&nlog, 4));
  CK(hipMalloc(&sink, 64));
  void* scratch;
  const size_t scratch_bytes = 1 << 20;
  CK(hipMalloc(&scratch, scratch_bytes));

  // real kernels: fixed finite inputs
  float *G = nullptr, *whh = nullptr, *Y = nullptr;
  std::vector<float> yref, ycur;
  lstm_single_fn f_single = nullptr;
  lstm_coop_fn f_coop = nullptr;
  scratch_fn f_scr = nullptr;
  status_fn f_status = nullptr;
  int64_t coop_scratch = 0;
  const int64_t g_elems = (int64_t)B * 8 * LH * N, y_elems = (int64_t)B * 2 * LH * N;
  bool want_real = false;
  for (auto& v : victims) want_real |= v.rfind("real", 0) == 0;
  if (want_real) {
    const char* path = getenv("ST2_LIB") ? getenv("ST2_LIB") : "styletts2_amd/libst2_hip.so";
    void* h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!h) {
      fprintf(stderr, "dlopen %s: %s (real_* victims skipped)\n", path, dlerror());
      victims.erase(std::remove_if(victims.begin(), victims.end(), [](const std::string& v) { return v.rfind("real", 0) == 0; }),
                    victims.end());
    } else {
      f_single = (lstm_single_fn)dlsym(h, "st2_lstm_bidir");
      f_coop = (lstm_coop_fn)dlsym(h, "st2_lstm_bidir_coop_recovering");
      f_scr = (scratch_fn)dlsym(h, "st2_lstm_coop_scratch_bytes");
      f_status = (status_fn)dlsym(h, "st2_status");
      if (!f_single || !f_coop || !f_scr) { fprintf(stderr, "missing LSTM symbols in %s\n", path); return 1; }
      coop_scratch = f_scr(B);
      CK(hipMalloc(&G, g_elems * 4));
      CK(hipMalloc(&whh, WN * 4 + 65536));
      CK(hipMalloc(&Y, y_elems * 4));
      std::vector<float> hg(g_elems), hw(WN);
      uint32_t st = 12345u;
      auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((int)(st >> 8 & 0xffff) - 32768) / 32768.f; };
      for (auto& v : hg) v = 2.f * rnd();
      for (auto& v : hw) v = rnd() / 8.f;
      CK(hipMemcpy(G, hg.data(), g_elems * 4, hipMemcpyHostToDevice));
      CK(hipMemcpy(whh, hw.data(), WN * 4, hipMemcpyHostToDevice));
      yref.resize(y_elems);
      ycur.resize(y_elems);
    }
  }

  // aggressor state
  std::vector<ConvLoad> convs;
  auto conv_of = [&](int ks, int cols) -> ConvLoad& {
    for (auto& c : convs)
      if (c.ks == ks && c.cols == cols) return c;
    convs.push_back(make_conv(ks, cols, 256, 5680, 1));
    return convs.back();
  };
  uint4* big;
  const size_t big_n = (size_t)64 << 20 >> 4;  // 64 MB
  CK(hipMalloc(&big, big_n * 16));
  CK(hipMemset(big, 1, big_n * 16));
  float* mf_out;
  CK(hipMalloc(&mf_out, 4096));

  auto launch_aggr = [&](const std::string& a, hipStream_t s) -> int {
    if (a == "none") return 0;
    if (a.rfind("xs_k", 0) == 0) {
      int ks = 0, cols = 0;
      sscanf(a.c_str(), "xs_k%d_n%d", &ks, &cols);
      return launch_conv(conv_of(ks, cols), s);
    }
    if (a == "empty") hipLaunchKernelGGL(aggr_empty, dim3(180), dim3(256), 0, s);
    else if (a == "samead") hipLaunchKernelGGL(aggr_samead, dim3(360), dim3(256), 0, s, big, big + 1, 400, 4096);
    else if (a == "mfma") hipLaunchKernelGGL(aggr_mfma, dim3(360), dim3(256), 0, s, mf_out, 1500);
    else if (a == "stream") hipLaunchKernelGGL(aggr_stream, dim3(180), dim3(256), 0, s, big, (size_t)4 << 20 >> 4, big + 2);
    else { fprintf(stderr, "unknown aggressor %s\n", a.c_str()); exit(1); }
    return 0;
  };
  auto launch_victim = [&](const std::string& v, int call, hipStream_t s) {
    if (v == "synth_single") {
      hipLaunchKernelGGL(victim_single, dim3(B, 2), dim3(256), 0, s, W, N, call, log, nlog, maxlog, sink);
    } else if (v.rfind("synth_coop", 0) == 0) {
      const int nblk = B <= 1 ? 1 : (B + 3) / 4;
      hipLaunchKernelGGL(zero16_kernel, dim3(18), dim3(256), 0, s, reinterpret_cast<uint4*>(scratch), (size_t)4352);
      if (v == "synth_coop_sc1")
        hipLaunchKernelGGL((victim_coop<1>), dim3(8, nblk, 2), dim3(256), 0, s, W, call, 0, 70, log, nlog, maxlog, sink);
      else
        hipLaunchKernelGGL((victim_coop<0>), dim3(8, nblk, 2), dim3(256), 0, s, W, call, v == "synth_coop_late" ? 20 : 0, 70, log, nlog,
                           maxlog, sink);
    } else if (v == "real_single") {
      if (f_single(G, (int64_t)8 * LH * N, N, whh, nullptr, B, LH, N, Y, (int64_t)2 * LH * N, N, s)) exit(2);
    } else if (v == "real_coop") {
      if (f_coop(G, (int64_t)8 * LH * N, N, whh, nullptr, B, LH, N, Y, (int64_t)2 * LH * N, N, scratch, coop_scratch, s)) exit(2);
    }
  };

  hipEvent_t ea0, ea1, eb0, eb1;
  CK(hipEventCreate(&ea0)); CK(hipEventCreate(&ea1)); CK(hipEventCreate(&eb0)); CK(hipEventCreate(&eb1));

  for (auto& v : victims) {
    const bool real = v.rfind("real", 0) == 0;
    // idle timing of the victim (and the reference output of the real kernels)
    for (int i = 0; i < 3; ++i) launch_victim(v, -1, sa);
    CK(hipStreamSynchronize(sa));
    CK(hipEventRecord(ea0, sa));
    for (int i = 0; i < 20; ++i) launch_victim(v, -1, sa);
    CK(hipEventRecord(ea1, sa));
    CK(hipEventSynchronize(ea1));
    float tv = 0;
    CK(hipEventElapsedTime(&tv, ea0, ea1));
    tv /= 20;
    if (real) CK(hipMemcpy(yref.data(), Y, y_elems * 4, hipMemcpyDeviceToHost));
    for (auto& a : aggrs) {
      float ta = 0.02f;
      if (a != "none") {
        for (int i = 0; i < 3; ++i) launch_aggr(a, sb);
        CK(hipStreamSynchronize(sb));
        CK(hipEventRecord(eb0, sb));
        for (int i = 0; i < 20; ++i) launch_aggr(a, sb);
        CK(hipEventRecord(eb1, sb));
        CK(hipEventSynchronize(eb1));
        CK(hipEventElapsedTime(&ta, eb0, eb1));
        ta /= 20;
      }
      int bad_calls = 0, total_calls = 0, bad_loads = 0;
      float t_a = 0, t_b = 0;
      std::vector<LogRec> recs;
      std::string detail;
      for (int trial = 0; trial < trials; ++trial) {
        CK(hipMemset(nlog, 0, 4));
        CK(hipDeviceSynchronize());
        // real kernels are compared call by call: batches of 10 calls into the same Y would hide all but the last, so the
        // real victims run `calls` single calls with a read-back each, the aggressor queue kept full around them
        if (!real) {
          const int n_aggr = a == "none" ? 0 : (int)(1.6f * calls * tv / ta) + 8;
          CK(hipEventRecord(eb0, sb));
          for (int i = 0; i < n_aggr; ++i) launch_aggr(a, sb);
          CK(hipEventRecord(eb1, sb));
          CK(hipEventRecord(ea0, sa));
          for (int i = 0; i < calls; ++i) launch_victim(v, trial * calls + i, sa);
          CK(hipEventRecord(ea1, sa));
          CK(hipDeviceSynchronize());
          float x = 0, y = 0;
          CK(hipEventElapsedTime(&x, ea0, ea1));
          CK(hipEventElapsedTime(&y, eb0, eb1));
          t_a += x; t_b += y;
          int n = 0;
          CK(hipMemcpy(&n, nlog, 4, hipMemcpyDeviceToHost));
          bad_loads += n;
          n = std::min(n, maxlog);
          std::vector<LogRec> r(n);
          if (n) CK(hipMemcpy(r.data(), log, sizeof(LogRec) * n, hipMemcpyDeviceToHost));
          std::vector<int> seen;
          for (auto& x2 : r) {
            if (std::find(seen.begin(), seen.end(), (int)x2.call) == seen.end()) seen.push_back((int)x2.call);
            recs.push_back(x2);
          }
          bad_calls += (int)seen.size();
          total_calls += calls;
        } else {
          for (int i = 0; i < calls; ++i) {
            const int n_aggr = a == "none" ? 0 : (int)(2.0f * tv / ta) + 6;
            for (int k = 0; k < n_aggr; ++k) launch_aggr(a, sb);
            launch_victim(v, i, sa);
            CK(hipStreamSynchronize(sa));
            CK(hipMemcpy(ycur.data(), Y, y_elems * 4, hipMemcpyDeviceToHost));
            if (memcmp(ycur.data(), yref.data(), y_elems * 4) != 0) {
              ++bad_calls;
              if (detail.size() < 1500) {  // where: earliest differing step per direction, the hidden units there
                char buf[256];
                for (int dir = 0; dir < 2; ++dir) {
                  int first_t = -1;
                  for (int s = 0; s < N && first_t < 0; ++s) {
                    const int t = dir == 0 ? s : N - 1 - s;
                    for (int u = 0; u < LH; ++u)
                      if (memcmp(&ycur[(size_t)(dir * LH + u) * N + t], &yref[(size_t)(dir * LH + u) * N + t], 4)) { first_t = t; break; }
                  }
                  if (first_t < 0) continue;
                  int lo = -1, hi = -1, cnt = 0;
                  for (int u = 0; u < LH; ++u)
                    if (memcmp(&ycur[(size_t)(dir * LH + u) * N + first_t], &yref[(size_t)(dir * LH + u) * N + first_t], 4)) {
                      if (lo < 0) lo = u;
                      hi = u;
                      ++cnt;
                    }
                  snprintf(buf, sizeof(buf), "    call %d dir %d: first differing t = %d, %d units in [%d, %d]\n", i, dir, first_t, cnt, lo, hi);
                  detail += buf;
                }
              }
            }
            ++total_calls;
          }
          CK(hipDeviceSynchronize());
        }
      }
      printf("victim=%-16s aggr=%-11s calls=%5d bad_calls=%4d bad_loads=%6d", v.c_str(), a.c_str(), total_calls, bad_calls, bad_loads);
      if (!real) printf("  victim %.2f ms/trial (idle %.3f ms/call), aggressor %.2f ms/trial (%.1f us/launch idle)", t_a / trials, tv, t_b / trials, ta * 1e3f);
      printf("\n");
      if (real && f_status) {
        const int st = f_status(1);
        if (st) printf("    st2_status = 0x%x\n", st);
      }
      if (!detail.empty()) fputs(detail.c_str(), stdout);
      // summarise the logged loads: runs of consecutive threads per (call, step, wg, k, gate)
      std::sort(recs.begin(), recs.end(), [](const LogRec& x, const LogRec& y) {
        if (x.call != y.call) return x.call < y.call;
        if (x.step != y.step) return x.step < y.step;
        if (x.wg != y.wg) return x.wg < y.wg;
        if (x.k != y.k) return x.k < y.k;
        if (x.gate != y.gate) return x.gate < y.gate;
        return x.tid < y.tid;
      });
      int printed = 0;
      for (size_t i = 0; i < recs.size() && printed < 12;) {
        size_t j = i + 1;
        while (j < recs.size() && recs[j].call == recs[i].call && recs[j].step == recs[i].step && recs[j].wg == recs[i].wg &&
               recs[j].k == recs[i].k && recs[j].gate == recs[i].gate && recs[j].tid == recs[j - 1].tid + 1)
          ++j;
        const LogRec& r = recs[i];
        // is the wrong word another element of the same buffer?  (i' = (got - WADD) * WMUL^-1 mod 2^32)
        uint32_t inv = 1;
        for (int it = 0; it < 5; ++it) inv *= 2u - WMUL * inv;
        const uint32_t src = (r.got - WADD) * inv;
        const uint32_t want_idx = (r.want - WADD) * inv;
        printf("    call %u step %u wg %u k %u gate %u threads %u..%u (%zu lanes) +%u us  got %08x want %08x", r.call, r.step, r.wg, r.k,
               r.gate, r.tid, recs[j - 1].tid, j - i, r.us_since_start, r.got, r.want);
        if (src < WN) printf(" = W[%u] (wanted W[%u], delta %d words)", src, want_idx, (int)src - (int)want_idx);
        printf("  reread %s / sc1 %s  hw_id %08x xcc %u\n", r.reread == r.want ? "ok" : "BAD", r.reread_sc1 == r.want ? "ok" : "BAD",
               r.hwid, r.xcc & 0xf);
        ++printed;
        i = j;
      }
      fflush(stdout);
    }
  }
  return 0;
}
