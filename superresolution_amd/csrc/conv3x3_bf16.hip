// The direct 3x3 convolution kernel's instantiations for SRMI_BF16 compute (conv3x3_kernel.h).
#include "conv3x3_kernel.h"

int srmi_conv3x3_launch_bf16(const srmi::ConvParams& p, int mb, int nb, hipStream_t s) { return srmi::launch_dt<SRMI_BF16>(p, mb, nb, s); }
int srmi_conv3x3_configure_bf16() { return srmi::configure_dt<SRMI_BF16>(); }
