// Explicit instantiations of the warp-specialised f16s conv for kernel size 7 (one translation unit per size for build time).
#include "st2_conv1d_f16s_ws.h"

template int st2ws::launch_ws_by_cout<7, 16>(const st2_conv_desc&, hipStream_t);
