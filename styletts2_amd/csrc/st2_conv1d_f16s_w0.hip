// Explicit instantiations of the warp-specialised f16s conv for kernel size 3 (one translation unit per size for build time).
#include "st2_conv1d_f16s_ws.h"

template int st2ws::launch_ws_by_cout<3, 32>(const st2_conv_desc&, hipStream_t);
