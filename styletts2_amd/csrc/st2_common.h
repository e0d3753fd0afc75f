// Shared helpers for the libst2_hip kernels (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include "st2.h"

void st2_set_error(const char* fmt, ...);
int st2_stream_cu_count(void* stream);  // st2_api.hip: CUs of a stream made by st2_stream_create_cu_mask, 0 otherwise
// st2_actsplit.hip: the engine's conv() names the conv site (st2_calibration_read's index) of the launches it is about to issue on
// this thread, so that the headroom records (st2_debug_headroom) can be folded into per-site operand scales; (null, -1) = none.
void st2_headroom_set_site(const void* engine, int site);
int* st2_status_device_ptr();  // st2_api.hip: device view of the sticky status word (nullptr without a device)

// Kernel side: raise status bit `bit` (ST2_STATUS_*).  Every bit has its own 32-bit slot in the host-mapped block so
// that raising is a plain system-scope STORE (no read-modify-write over the host link); st2_status() ORs the slots.
__device__ __forceinline__ void st2_raise_status(int* status, int bit) {
  if (status) __hip_atomic_store(status + (bit == ST2_STATUS_F16_RANGE ? 0 : (bit == ST2_STATUS_LSTM_TIMEOUT ? 1 :
                                           (bit == ST2_STATUS_DURATION_SUM ? 2 : 3))), bit,
                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

#define ST2_REQUIRE(cond, ...)            \
  do {                                    \
    if (!(cond)) {                        \
      st2_set_error(__VA_ARGS__);         \
      return 1;                           \
    }                                     \
  } while (0)

#define ST2_CHECK_LAUNCH(name)                                   \
  do {                                                           \
    hipError_t e__ = hipGetLastError();                          \
    if (e__ != hipSuccess) {                                     \
      st2_set_error("%s: %s", name, hipGetErrorString(e__));     \
      return 1;                                                  \
    }                                                            \
  } while (0)

static inline int st2_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// hipFuncSetAttribute (dynamic LDS above 64 KB) is a PER-DEVICE setting: every launcher keeps one bit per device ordinal
// and kernel instantiation, so that a process driving several GPUs sets it on each of them (advisor, round 3).  The bit is
// published only AFTER `set_attr` has returned (advisor, round 4): a second thread that finds it clear sets the attribute
// again (harmless) instead of launching a > 64 KB kernel while the first thread is still inside hipFuncSetAttribute.
// Ordinals >= 64 set it on every launch (legal).
template <class F>
static inline void st2_once_per_device(std::atomic<uint64_t>& mask, F&& set_attr) {
  int dev = 0;
  const bool known = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64;
  if (known && (mask.load(std::memory_order_acquire) & (1ull << dev))) return;
  set_attr();
  if (known) mask.fetch_or(1ull << dev, std::memory_order_release);
}

// 64-lane butterfly-free tree (fixed order => bitwise reproducible).
__device__ __forceinline__ double st2_wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
__device__ __forceinline__ float st2_wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
