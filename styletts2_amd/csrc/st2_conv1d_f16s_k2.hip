// Explicit instantiations of the f16s conv for kernel sizes [7, 11] (split over translation units for build time).
#include "st2_conv1d_f16s_impl.h"

template int st2f16s::launch_by_cout<7, 16>(const st2_conv_desc&, hipStream_t);
template int st2f16s::launch_by_cout<11, 16>(const st2_conv_desc&, hipStream_t);
