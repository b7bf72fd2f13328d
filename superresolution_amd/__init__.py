"""superresolution_amd — MI355X (gfx950) native inference for EDSR, RCAN and Real-ESRGAN RRDBNet.

The Python side keeps the reference's module surface (constructors, state_dict key layout,
forward(x)); the arithmetic runs in hand-written HIP kernels behind the C ABI of
include/srmi355.h.  There is no CPU execution path.
"""
from .edsr_model import EDSR, EDSRCore, load_edsr_from_checkpoint  # noqa: F401
from .rcan_model import RCAN, RCANCore, load_rcan_from_checkpoint  # noqa: F401
from .realesrgan_model import RRDBNet, load_realesrgan_from_checkpoint  # noqa: F401

__all__ = ["EDSR", "EDSRCore", "RCAN", "RCANCore", "RRDBNet", "load_edsr_from_checkpoint",
           "load_rcan_from_checkpoint", "load_realesrgan_from_checkpoint"]
