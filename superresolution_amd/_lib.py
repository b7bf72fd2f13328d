"""ctypes binding of libsrmi355.so (the C ABI declared in include/srmi355.h).

There is no fallback: if the shared library is missing, import of this module raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SRMI_LIB points diagnostic runs at another build of the same library (A/B timing of kernel variants)
LIB_PATH = os.environ.get("SRMI_LIB") or os.path.join(_HERE, "libsrmi355.so")

F32, F16, BF16, I8 = 0, 1, 2, 3
ACT_NONE, ACT_RELU, ACT_LRELU = 0, 1, 2
ROUND_HALF_EVEN, ROUND_HALF_UP, ROUND_TRUNC = 0, 1, 2
IN_F32_NCHW, IN_U8_NHWC = 0, 1
TAIL_NONE, TAIL_F32_NCHW, TAIL_U8_NHWC = 0, 1, 2
ARCH_EDSR, ARCH_RCAN, ARCH_RRDBNET = 1, 2, 3

DTYPE_NAMES = {"f32": F32, "fp32": F32, "float32": F32, "f16": F16, "fp16": F16, "float16": F16,
               "half": F16, "bf16": BF16, "bfloat16": BF16}


class SrmiError(RuntimeError):
    pass


class Tensor(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("img_stride", C.c_int64), ("plane_stride", C.c_int64),
                ("dtype", C.c_int32), ("flags", C.c_int32), ("qscale", C.c_float), ("reserved", C.c_int32)]


TENSOR_ROUND8 = 1
ALGO_DIRECT, ALGO_WINOGRAD, ALGO_WINOGRAD_X, ALGO_SUBPIXEL_UP2 = 0, 1, 2, 3
CA_SLOTS = 8   # SRMI_CA_SLOTS


class ConvDesc(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32),
                ("cin", C.c_int32), ("cout", C.c_int32), ("in_up2", C.c_int32), ("act", C.c_int32),
                ("ps_r", C.c_int32), ("cout_tile", C.c_int32), ("rows_per_wave", C.c_int32),
                ("tail_mode", C.c_int32),
                ("inp", Tensor), ("wpk", C.c_void_p), ("bias", C.c_void_p),
                ("alpha", C.c_float), ("beta", C.c_float),
                ("res1", Tensor), ("res2", Tensor), ("out", Tensor), ("out2", Tensor),
                ("chansum", C.c_void_p), ("tail_out", C.c_void_p),
                ("tail_mat", C.c_float * 9), ("tail_bias", C.c_float * 3),
                ("tail_lo", C.c_float), ("tail_hi", C.c_float), ("tail_scale", C.c_float),
                ("tail_round", C.c_int32), ("wscale", C.c_void_p),
                ("in_gate", Tensor), ("gate", C.c_void_p), ("gate_alpha", C.c_float), ("algo", C.c_int32),
                ("in_sum", Tensor),
                ("ca_sums", C.c_void_p), ("ca_count", C.c_void_p), ("ca_w1", C.c_void_p), ("ca_b1", C.c_void_p),
                ("ca_w2", C.c_void_p), ("ca_b2", C.c_void_p), ("ca_scale", C.c_void_p), ("ca_cr", C.c_int32),
                ("ca_reserved", C.c_int32)]


class RcabDesc(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("c", C.c_int32),
                ("gate_alpha", C.c_float),
                ("inp", Tensor), ("in_gate", Tensor), ("in_sum", Tensor),
                ("gate", C.c_void_p), ("w1", C.c_void_p), ("b1", C.c_void_p), ("w2", C.c_void_p), ("b2", C.c_void_p),
                ("out", Tensor),
                ("ca_sums", C.c_void_p), ("ca_count", C.c_void_p), ("ca_w1", C.c_void_p), ("ca_b1", C.c_void_p),
                ("ca_w2", C.c_void_p), ("ca_b2", C.c_void_p), ("ca_scale", C.c_void_p), ("ca_cr", C.c_int32),
                ("ca_reserved", C.c_int32)]


class ConvFirstDesc(C.Structure):
    _fields_ = [("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("cout", C.c_int32),
                ("in_kind", C.c_int32), ("reserved", C.c_int32), ("inp", C.c_void_p),
                ("pre_mat", C.c_float * 9), ("pre_bias", C.c_float * 3),
                ("w_kcio", C.c_void_p), ("bias", C.c_void_p), ("out", Tensor), ("out2", Tensor)]


class ModelDesc(C.Structure):
    _fields_ = [("arch", C.c_int32), ("dtype", C.c_int32), ("scale", C.c_int32),
                ("n_feats", C.c_int32), ("n_blocks", C.c_int32), ("n_groups", C.c_int32),
                ("grow_ch", C.c_int32), ("reduction", C.c_int32), ("res_scale", C.c_float),
                ("with_meanshift", C.c_int32), ("trunk_f32", C.c_int32), ("device", C.c_int32)]


# every symbol include/srmi355.h (+ the runner headers) declares; tests check they all resolve
EXPORTS = [
    "srmi_last_error", "srmi_device_count", "srmi_version",
    "srmi_conv3x3", "srmi_conv3x3_tiles", "srmi_pack_conv3x3_bytes", "srmi_pack_conv3x3",
    "srmi_pack_wino3x3_bytes", "srmi_pack_wino3x3", "srmi_pack_wino1d3x3_bytes", "srmi_pack_wino1d3x3", "srmi_model_set_winograd", "srmi_model_get_winograd",
    "srmi_rcab_fused", "srmi_model_set_fused_rcab", "srmi_model_get_fused_rcab",
    "srmi_pack_conv3x3_up2_bytes", "srmi_pack_conv3x3_up2", "srmi_model_set_subpixel_up", "srmi_model_get_subpixel_up",
    "srmi_conv_first", "srmi_ca_scale", "srmi_ca_apply", "srmi_channel_sums",
    "srmi_pixel_shuffle", "srmi_nearest_up2", "srmi_nchw_to_planes", "srmi_planes_to_nchw",
    "srmi_resize_bilinear_u8",
    "srmi_model_create", "srmi_model_destroy", "srmi_model_set_tensor", "srmi_model_num_keys",
    "srmi_model_key", "srmi_model_finalize", "srmi_model_forward_f32", "srmi_model_forward_u8",
    "srmi_model_workspace_bytes", "srmi_model_set_sweep", "srmi_model_set_fused_attention", "srmi_model_set_lanes", "srmi_model_set_growth_round8", "srmi_model_flops", "srmi_model_set_rgb_shift",
    "srmi_absmax", "srmi_model_calibrate_begin", "srmi_model_calibrate_end", "srmi_model_set_w8a8",
    "srmi_model_calibration", "srmi_model_set_calibration", "srmi_model_set_calibration_ex", "srmi_model_get_w8a8", "srmi_model_reserve", "srmi_model_forward_u8_host", "srmi_model_reserve_host",
    "srmi_model_forward_u8_noalloc", "srmi_static_workspace_bytes",
    "edsr_init", "edsr_release", "edsr_inference", "edsr_inference_ex",
    "rcan_init", "rcan_release", "rcan_inference", "rcan_inference_ex",
    "realesrgan_init", "realesrgan_release", "realesrgan_inference", "realesrgan_inference_ex",
    "sr_init", "sr_release", "sr_inference", "sr_reserve", "sr_last_error",
    "srmi_executor_create", "srmi_executor_destroy", "srmi_executor_run", "srmi_executor_input_size",
    "srmi_executor_output_size",
]

_lib = None


def lib():
    """The loaded library (loads on first use; raises SrmiError if it is not built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SrmiError(f"{LIB_PATH} is not built: run `python -c 'import __graft_entry__ as g; "
                            "g.build()'` or `make -C superresolution_amd/csrc`")
        L = C.CDLL(LIB_PATH)
        L.srmi_last_error.restype = C.c_char_p
        L.srmi_version.restype = C.c_char_p
        L.srmi_pack_conv3x3_bytes.restype = C.c_size_t
        L.srmi_pack_conv3x3_bytes.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
        L.srmi_pack_conv3x3.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p]
        L.srmi_pack_wino3x3_bytes.restype = C.c_size_t
        L.srmi_pack_wino3x3_bytes.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int)]
        L.srmi_pack_wino3x3.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.srmi_pack_wino1d3x3_bytes.restype = C.c_size_t
        L.srmi_pack_wino1d3x3_bytes.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int)]
        L.srmi_pack_wino1d3x3.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.srmi_pack_conv3x3_up2_bytes.restype = C.c_size_t
        L.srmi_pack_conv3x3_up2_bytes.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
        L.srmi_pack_conv3x3_up2.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.srmi_model_set_subpixel_up.argtypes = [C.c_void_p, C.c_int]
        L.srmi_model_get_subpixel_up.argtypes = [C.c_void_p]
        L.srmi_model_set_winograd.argtypes = [C.c_void_p, C.c_int]
        L.srmi_model_get_winograd.argtypes = [C.c_void_p]
        L.srmi_rcab_fused.argtypes = [C.POINTER(RcabDesc), C.c_void_p]
        L.srmi_model_set_fused_rcab.argtypes = [C.c_void_p, C.c_int]
        L.srmi_model_get_fused_rcab.argtypes = [C.c_void_p]
        L.srmi_conv3x3.argtypes = [C.POINTER(ConvDesc), C.c_void_p]
        L.srmi_conv3x3_tiles.argtypes = [C.c_int, C.c_int, C.c_int]
        L.srmi_conv_first.argtypes = [C.POINTER(ConvFirstDesc), C.c_void_p]
        L.srmi_ca_scale.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p]
        L.srmi_ca_apply.argtypes = [C.POINTER(Tensor), C.POINTER(Tensor), C.c_void_p, C.c_float,
                                    C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Tensor),
                                    C.POINTER(Tensor), C.c_void_p]
        L.srmi_channel_sums.argtypes = [C.POINTER(Tensor), C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_void_p, C.c_void_p]
        L.srmi_pixel_shuffle.argtypes = [C.POINTER(Tensor), C.POINTER(Tensor), C.c_int, C.c_int,
                                         C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.srmi_nearest_up2.argtypes = [C.POINTER(Tensor), C.POINTER(Tensor), C.c_int, C.c_int,
                                       C.c_int, C.c_int, C.c_void_p]
        L.srmi_nchw_to_planes.argtypes = [C.c_void_p, C.POINTER(Tensor), C.c_int, C.c_int, C.c_int,
                                          C.c_int, C.c_void_p]
        L.srmi_planes_to_nchw.argtypes = [C.POINTER(Tensor), C.c_void_p, C.c_int, C.c_int, C.c_int,
                                          C.c_int, C.c_void_p]
        L.srmi_resize_bilinear_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                              C.c_int, C.c_void_p]
        L.srmi_model_create.argtypes = [C.POINTER(ModelDesc), C.POINTER(C.c_void_p)]
        L.srmi_model_destroy.argtypes = [C.c_void_p]
        L.srmi_model_destroy.restype = None
        L.srmi_model_set_tensor.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p,
                                            C.POINTER(C.c_int64), C.c_int]
        L.srmi_model_num_keys.argtypes = [C.c_void_p]
        L.srmi_model_key.argtypes = [C.c_void_p, C.c_int]
        L.srmi_model_key.restype = C.c_char_p
        L.srmi_model_finalize.argtypes = [C.c_void_p]
        L.srmi_model_forward_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                             C.c_int, C.c_void_p]
        L.srmi_model_forward_u8.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                            C.c_int, C.c_int, C.c_void_p]
        L.srmi_model_forward_u8_noalloc.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                                    C.c_int, C.c_int, C.c_void_p]
        L.srmi_model_forward_u8_host.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.srmi_model_reserve_host.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        if hasattr(L, "srmi_model_calibrate_begin"):  # (absent from older builds loaded through $SRMI_LIB)
            L.srmi_absmax.argtypes = [C.POINTER(Tensor), C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
            L.srmi_model_calibrate_begin.argtypes = [C.c_void_p]
            L.srmi_model_calibrate_end.argtypes = [C.c_void_p]
            L.srmi_model_set_w8a8.argtypes = [C.c_void_p, C.c_int]
            L.srmi_model_calibration.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
            L.srmi_model_set_calibration.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
            L.srmi_model_set_calibration_ex.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
            L.srmi_model_get_w8a8.argtypes = [C.c_void_p]
        L.srmi_model_reserve.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.srmi_static_workspace_bytes.argtypes = []
        L.srmi_static_workspace_bytes.restype = C.c_size_t
        L.srmi_model_workspace_bytes.argtypes = [C.c_void_p]
        L.srmi_model_workspace_bytes.restype = C.c_size_t
        L.srmi_model_set_sweep.argtypes = [C.c_void_p, C.c_int]
        if hasattr(L, "srmi_model_set_fused_attention"):
            L.srmi_model_set_fused_attention.argtypes = [C.c_void_p, C.c_int]
        L.srmi_model_set_lanes.argtypes = [C.c_void_p, C.c_int]
        L.srmi_model_set_growth_round8.argtypes = [C.c_void_p, C.c_int]
        L.srmi_model_flops.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.srmi_model_flops.restype = C.c_double
        L.srmi_model_set_rgb_shift.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        _lib = L
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().srmi_last_error()
        raise SrmiError(f"{what}: {msg.decode() if msg else 'unknown error'}")
