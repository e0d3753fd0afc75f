// Explicit instantiations of the warp-specialised f16s conv for kernel size 11 (one translation unit per size for build time).
#include "st2_conv1d_f16s_ws.h"

template int st2ws::launch_ws_by_cout<11, 16>(const st2_conv_desc&, hipStream_t);

#ifdef ST2_WS_TIMELINE
// measurement build: copies the phase stamps of workgroup 0 (12 waves x TL_N) of the LAST launch to the host and clears them
extern "C" int st2_debug_ws_timeline(unsigned long long* host) {
  unsigned long long* p = st2ws::tl_buffer();
  if (!p || hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpy(host, p, 12 * st2ws::TL_N * 8, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  (void)hipMemset(p, 0, 12 * st2ws::TL_N * 8);
  return st2ws::TL_N;
}
#endif
