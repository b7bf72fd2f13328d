// The direct 3x3 convolution kernel's instantiations for SRMI_F32 compute (conv3x3_kernel.h).
#include "conv3x3_kernel.h"

int srmi_conv3x3_launch_f32(const srmi::ConvParams& p, int mb, int nb, hipStream_t s) { return srmi::launch_dt<SRMI_F32>(p, mb, nb, s); }
int srmi_conv3x3_configure_f32() { return srmi::configure_dt<SRMI_F32>(); }
