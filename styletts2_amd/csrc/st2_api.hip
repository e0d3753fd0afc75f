// Library-level entry points: version, error string, device probe.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <map>
#include <mutex>
#include "st2_common.h"

static thread_local char g_err[512] = "";

void st2_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int st2_abi_version(void) { return ST2_ABI_VERSION; }
extern "C" const char* st2_last_error(void) { return g_err; }
extern "C" int st2_sizeof_conv_desc(void) { return (int)sizeof(st2_conv_desc); }

extern "C" int st2_device_info(int dev, char* name, int cap) {
  hipDeviceProp_t p;
  hipError_t e = hipGetDeviceProperties(&p, dev);
  if (e != hipSuccess) {
    st2_set_error("st2_device_info: %s", hipGetErrorString(e));
    return -1;
  }
  if (name && cap > 0) {
    strncpy(name, p.gcnArchName, cap - 1);
    name[cap - 1] = 0;
  }
  return p.multiProcessorCount;
}

// ---- sticky status word (host-mapped, see st2.h) --------------------------------------------------------------------
static int* g_status_host = nullptr;  // host view
static int* g_status_dev = nullptr;   // device view of the same 64 bytes
static bool g_status_tried = false;

// Device pointer the kernels OR their status bits into; nullptr when no device is usable (the kernels then skip the
// report).  First call allocates -- never inside a stream capture: every launcher that reports calls this before it
// launches, and graph users run one eager pass first (GraphedSampler / the engine's capture path do).
int* st2_status_device_ptr() {
  if (!g_status_tried) {
    g_status_tried = true;
    void* h = nullptr;
    if (hipHostMalloc(&h, 64, hipHostMallocMapped) == hipSuccess && h) {
      memset(h, 0, 64);
      void* dptr = nullptr;
      if (hipHostGetDevicePointer(&dptr, h, 0) == hipSuccess && dptr) {
        g_status_host = reinterpret_cast<int*>(h);
        g_status_dev = reinterpret_cast<int*>(dptr);
      } else {
        (void)hipHostFree(h);
      }
    }
    (void)hipGetLastError();
  }
  return g_status_dev;
}

extern "C" int st2_status(int clear) {
  if (!st2_status_device_ptr()) {
    st2_set_error("st2_status: no HIP device / host-mapped allocation failed");
    return -1;
  }
  int v = 0;
  for (int i = 0; i < 4; ++i)
    v |= clear ? __atomic_exchange_n(g_status_host + i, 0, __ATOMIC_SEQ_CST) : __atomic_load_n(g_status_host + i, __ATOMIC_SEQ_CST);
  return v;
}

// ---- CU-partitioned streams (st2.h, ABI v17) -----------------------------------------------------------------------
// Streams created with a CU mask, and how many CUs each owns: a cooperative launch (st2_lstm_bidir_coop) on such a stream
// must fit ITS CUs, not the device's (advisor, round 3: on a 16-64-CU partition the spin-waiting groups would otherwise be
// only partially resident and time out).
static std::mutex g_mask_mu;
static std::map<void*, int> g_masked_streams;

int st2_stream_cu_count(void* stream) {  // 0 = not a CU-masked stream of this library
  std::lock_guard<std::mutex> lock(g_mask_mu);
  auto it = g_masked_streams.find(stream);
  return it == g_masked_streams.end() ? 0 : it->second;
}

extern "C" int st2_stream_create_cu_mask(const uint32_t* mask, int32_t n_words, void** stream) {
  if (!mask || n_words <= 0 || !stream) {
    st2_set_error("st2_stream_create_cu_mask: bad arguments");
    return 1;
  }
  bool any = false;
  for (int i = 0; i < n_words; ++i) any |= mask[i] != 0;
  if (!any) {
    st2_set_error("st2_stream_create_cu_mask: empty CU mask");
    return 1;
  }
  hipStream_t s = nullptr;
  hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)n_words, mask);
  if (e != hipSuccess) {
    st2_set_error("st2_stream_create_cu_mask: %s", hipGetErrorString(e));
    (void)hipGetLastError();
    return 1;
  }
  *stream = s;
  int cus = 0;
  for (int i = 0; i < n_words; ++i) cus += __builtin_popcount(mask[i]);
  {
    std::lock_guard<std::mutex> lock(g_mask_mu);
    g_masked_streams[s] = cus;
  }
  return 0;
}

extern "C" int st2_stream_destroy(void* stream) {
  if (!stream) return 0;
  {
    std::lock_guard<std::mutex> lock(g_mask_mu);
    g_masked_streams.erase(stream);
  }
  hipError_t e = hipStreamDestroy(reinterpret_cast<hipStream_t>(stream));
  if (e != hipSuccess) {
    st2_set_error("st2_stream_destroy: %s", hipGetErrorString(e));
    (void)hipGetLastError();
    return 1;
  }
  return 0;
}
