// Explicit instantiations of the xs conv for kernel sizes [1, 2] (split over translation units for build time).
#include "st2_conv1d_xs_impl.h"

template int st2xs::launch_by_cout<1, 32>(const st2_conv_desc&, hipStream_t, int);
template int st2xs::launch_by_cout<2, 32>(const st2_conv_desc&, hipStream_t, int);
