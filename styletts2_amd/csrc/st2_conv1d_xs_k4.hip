// Explicit instantiation of the xs conv for kernel size 3 on 16-channel chunks (variant XS_V_CHUNK16; own translation
// unit for build time).
#include "st2_conv1d_xs_impl.h"

template int st2xs::launch_by_cout<3, 16>(const st2_conv_desc&, hipStream_t, int);
