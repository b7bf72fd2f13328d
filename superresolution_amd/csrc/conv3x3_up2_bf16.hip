// The sub-pixel (2x2 on the source grid) form of "nearest-x2 upsampling, then 3x3 convolution" for SRMI_BF16 compute: the
// KT == 2 instantiations of conv3x3_kernel.h.
#include "conv3x3_kernel.h"

int srmi_conv3x3_launch_up2_bf16(const srmi::ConvParams& p, int mb, int nb, hipStream_t s) { return srmi::launch_up2_dt<SRMI_BF16>(p, mb, nb, s); }
int srmi_conv3x3_configure_up2_bf16() { return srmi::configure_up2_dt<SRMI_BF16>(); }
