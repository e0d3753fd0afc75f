// Library-level entry points: version, error string, device probe.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
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
