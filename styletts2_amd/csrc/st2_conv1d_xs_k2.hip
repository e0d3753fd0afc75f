// Explicit instantiations of the xs conv for kernel size 7 (split over translation units for build time).
#include "st2_conv1d_xs_impl.h"

template int st2xs::launch_by_cout<7, 16>(const st2_conv_desc&, hipStream_t, int);
