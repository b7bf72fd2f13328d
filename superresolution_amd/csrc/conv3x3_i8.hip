// The direct 3x3 convolution kernel's instantiations for SRMI_I8 compute (conv3x3_kernel.h).
#include "conv3x3_kernel.h"

int srmi_conv3x3_launch_i8(const srmi::ConvParams& p, int mb, int nb, hipStream_t s) { return srmi::launch_dt<SRMI_I8>(p, mb, nb, s); }
int srmi_conv3x3_configure_i8() { return srmi::configure_dt<SRMI_I8>(); }
