// Explicit instantiations of the f16s conv for kernel sizes [1, 2] (split over translation units for build time).
#include "st2_conv1d_f16s_impl.h"

template int st2f16s::launch_by_cout<1, 32>(const st2_conv_desc&, hipStream_t);
template int st2f16s::launch_by_cout<2, 32>(const st2_conv_desc&, hipStream_t);
