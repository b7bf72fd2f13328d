"""Model-level Python surface: `Engine` wraps the C executor (srmi_model_* in include/srmi355.h),
`HipModule` is the nn.Module base the three network classes derive from.

A forward pass is ONE ctypes call; all layer sequencing, workspace management and kernel
launches happen inside libsrmi355.so on the caller's current HIP stream.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from . import custom_ops  # noqa: F401  (registers torch.ops.srmi355.*)
from ._lib import check, lib


def resolve_dtype(name):
    """'f32' | 'f16' | 'bf16' (or None -> $SRMI_DTYPE, default f16) -> library dtype code."""
    if name is None:
        name = os.environ.get("SRMI_DTYPE", "f16")
    if isinstance(name, torch.dtype):
        name = {torch.float32: "f32", torch.float16: "f16", torch.bfloat16: "bf16"}[name]
    if isinstance(name, str):
        try:
            return _lib.DTYPE_NAMES[name.lower()]
        except KeyError:
            raise ValueError(f"unknown compute dtype {name!r}") from None
    return int(name)


class Engine:
    """Owns one `srmi_model_t`."""

    def __init__(self, arch, *, dtype, scale, n_feats, n_blocks, n_groups=0, grow_ch=0, reduction=0,
                 res_scale=1.0, with_meanshift=False, trunk_f32=True, device=0, growth_round8=True):
        d = _lib.ModelDesc()
        d.arch, d.dtype, d.scale = arch, dtype, scale
        d.n_feats, d.n_blocks, d.n_groups = n_feats, n_blocks, n_groups
        d.grow_ch, d.reduction, d.res_scale = grow_ch, reduction, res_scale
        d.with_meanshift, d.trunk_f32, d.device = int(with_meanshift), int(trunk_f32), device
        self.desc = d
        self.scale = scale
        self.device = torch.device("cuda", device)
        h = C.c_void_p()
        # the destructor may run while the interpreter shuts down, when module globals are already gone:
        # keep the loaded library on the instance
        self._destroy = lib().srmi_model_destroy
        check(lib().srmi_model_create(C.byref(d), C.byref(h)), "srmi_model_create")
        self._h = h
        self._finalized = False
        if not growth_round8:
            check(lib().srmi_model_set_growth_round8(h, 0), "srmi_model_set_growth_round8")

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._destroy(h)

    __del__ = close

    def keys(self):
        n = lib().srmi_model_num_keys(self._h)
        return [lib().srmi_model_key(self._h, i).decode() for i in range(n)]

    def load_state_dict(self, sd):
        """sd: mapping key -> float tensor / ndarray in the reference's layout (strict)."""
        for k, v in sd.items():
            if isinstance(v, torch.Tensor):
                v = v.detach().to("cpu", torch.float32).contiguous().numpy()
            a = np.ascontiguousarray(np.asarray(v, dtype=np.float32))
            shape = (C.c_int64 * a.ndim)(*a.shape)
            check(lib().srmi_model_set_tensor(self._h, k.encode(), a.ctypes.data, shape, a.ndim),
                  "srmi_model_set_tensor")
        check(lib().srmi_model_finalize(self._h), "srmi_model_finalize")
        self._finalized = True

    def set_fused_attention(self, on):
        """RCAN, half-precision trunk: whether the consuming convolution forms a channel-attention block's output
        while loading (default) or a separate elementwise pass does; the results are the same bytes."""
        check(lib().srmi_model_set_fused_attention(self._h, int(bool(on))), "srmi_model_set_fused_attention")

    def set_fused_rcab(self, on=True):
        """RCAN and EDSR with f16 operands and 64 features (trunk in f16 or f32): both convolutions of a block in one
        kernel (True) or as two launches (False)."""
        check(lib().srmi_model_set_fused_rcab(self._h, int(bool(on))), "srmi_model_set_fused_rcab")

    def fused_rcab(self):
        return bool(lib().srmi_model_get_fused_rcab(self._h))

    def set_subpixel_up(self, on=True):
        """RRDBNet, f16 compute: conv_up1 / conv_up2 (behind the nearest-x2 upsamplings) as four 2x2 convolutions on
        the source grid (True: 4/9 of the multiplies) or as nine-tap convolutions reading through the x2 address map."""
        check(lib().srmi_model_set_subpixel_up(self._h, int(bool(on))), "srmi_model_set_subpixel_up")

    def subpixel_up(self):
        return bool(lib().srmi_model_get_subpixel_up(self._h))

    def set_winograd(self, on=True):
        """RRDBNet, f16 compute: run the dense blocks' convolutions in the direct form (False / 0), the Winograd
        F(2x2,3x3) form (True / 1) or the Winograd F(2,3)-along-x form (2); takes effect at the next forward."""
        check(lib().srmi_model_set_winograd(self._h, int(on)), "srmi_model_set_winograd")

    def winograd(self):
        """0 direct, 1 F(2x2,3x3), 2 F(2,3) along x."""
        return int(lib().srmi_model_get_winograd(self._h))

    def set_sweep(self, images_per_sweep):
        check(lib().srmi_model_set_sweep(self._h, int(images_per_sweep)), "srmi_model_set_sweep")

    def set_lanes(self, lanes):
        """Internal HIP streams (1..4) the sweeps of one call are spread over; results do not depend on it."""
        check(lib().srmi_model_set_lanes(self._h, int(lanes)), "srmi_model_set_lanes")

    def set_rgb_shift(self, shift):
        """Core (no-MeanShift) EDSR / RCAN on the u8 path: subtract shift[c] before the first conv and add it back
        after the last (the runners' rgb_mean * rgb_range, edsr.cpp:36-37,116)."""
        v = (C.c_float * 3)(*[float(x) for x in shift])
        check(lib().srmi_model_set_rgb_shift(self._h, v), "srmi_model_set_rgb_shift")

    def calibrate(self, frames, batch=4):
        """w8a8 calibration (RRDBNet, f16 compute): run `frames` (uint8 NHWC, or float32 NCHW in [0,1] as the
        reference's calibration .npy files hold them, realesrgan/rknn/python/convert.py:198-213) through the network
        while the activation ranges of the dense blocks are recorded, then switch the int8 mode on.  Returns the
        recorded max |x| values."""
        check(lib().srmi_model_calibrate_begin(self._h), "srmi_model_calibrate_begin")
        try:
            t = torch.as_tensor(frames)
            for i in range(0, t.shape[0], batch):
                part = t[i:i + batch].to(self.device)
                if part.dtype == torch.uint8:
                    self.forward_u8(part)
                else:
                    self.forward_f32(part)
            torch.cuda.synchronize(self.device)
        finally:
            rc = lib().srmi_model_calibrate_end(self._h)
        check(rc, "srmi_model_calibrate_end")
        n = lib().srmi_model_calibration(self._h, None, 0)
        a = np.zeros(n, dtype=np.float32)
        lib().srmi_model_calibration(self._h, a.ctypes.data, n)
        return a

    def calibration(self):
        """max |x| of the calibrated tensors (empty array when the model is not calibrated)."""
        n = lib().srmi_model_calibration(self._h, None, 0)
        a = np.zeros(n, dtype=np.float32)
        if n:
            lib().srmi_model_calibration(self._h, a.ctypes.data, n)
        return a

    def set_calibration(self, absmax, w8a8=True):
        """Install recorded activation ranges (see calibration()); the int8 mode is switched on unless w8a8=False
        (calibrated, but running in f16 until set_w8a8(True))."""
        a = np.ascontiguousarray(np.asarray(absmax, dtype=np.float32))
        check(lib().srmi_model_set_calibration_ex(self._h, a.ctypes.data, a.size, int(bool(w8a8))),
              "srmi_model_set_calibration")

    def w8a8(self):
        """True while the int8 dense blocks are the form that runs."""
        return bool(lib().srmi_model_get_w8a8(self._h))

    def set_w8a8(self, on=True):
        """Switch a calibrated model between its int8 dense blocks and plain f16."""
        check(lib().srmi_model_set_w8a8(self._h, int(bool(on))), "srmi_model_set_w8a8")

    def workspace_bytes(self):
        return int(lib().srmi_model_workspace_bytes(self._h))

    def flops(self, h, w):
        return float(lib().srmi_model_flops(self._h, h, w))

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _check_out(self, out, shape, dtype):
        """A caller-supplied result tensor is handed to the kernels as a raw pointer: it must be exactly the
        dense tensor they write."""
        if (out.dtype != dtype or out.device != self.device or tuple(out.shape) != tuple(shape)
                or not out.is_contiguous()):
            raise ValueError(f"out must be a contiguous {dtype} tensor of shape {tuple(shape)} on {self.device}, got "
                             f"{out.dtype} {tuple(out.shape)} on {out.device}"
                             f"{'' if out.is_contiguous() else ' (not contiguous)'}")
        return out

    def reserve(self, n, h, w, host=False):
        """Allocate everything a later forward of up to n frames of h x w needs (workspace, lanes; with
        host=True also the pinned staging of forward_u8_host)."""
        fn = lib().srmi_model_reserve_host if host else lib().srmi_model_reserve
        check(fn(self._h, int(n), int(h), int(w)), "srmi_model_reserve")

    def forward_u8_host(self, imgs: np.ndarray, round_mode=_lib.ROUND_HALF_UP, out: np.ndarray = None) -> np.ndarray:
        """uint8 NHWC frames in host memory -> uint8 NHWC frames in host memory; transfers overlap compute."""
        a = np.ascontiguousarray(imgs)
        if a.ndim != 4 or a.shape[3] != 3 or a.dtype != np.uint8:
            raise ValueError(f"expected uint8 [N,H,W,3], got {a.dtype} {a.shape}")
        n, h, w, _ = a.shape
        shape = (n, h * self.scale, w * self.scale, 3)
        if out is None:
            out = np.empty(shape, dtype=np.uint8)
        elif out.dtype != np.uint8 or out.shape != shape or not out.flags.c_contiguous:
            raise ValueError(f"out must be a C-contiguous uint8 array of shape {shape}")
        if n == 0:
            return out
        check(lib().srmi_model_forward_u8_host(self._h, a.ctypes.data, out.ctypes.data, n, h, w, round_mode),
              "srmi_model_forward_u8_host")
        return out

    def handle(self) -> int:
        """The srmi_model_t* as an integer: what torch.ops.srmi355.forward_f32 / forward_u8 take (custom_ops.py)."""
        return int(self._h.value)

    def forward_f32(self, x: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
        """x: float32 NCHW on this engine's GPU -> float32 NCHW [n,3,h*scale,w*scale].  Without `out` the call goes
        through the registered operator torch.ops.srmi355.forward_f32 (same C entry point underneath)."""
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError(f"expected [N,3,H,W], got {tuple(x.shape)}")
        x = x.to(self.device, torch.float32).contiguous()
        if out is None:
            return torch.ops.srmi355.forward_f32(self.handle(), self.scale, x)
        n, _, h, w = x.shape
        shape = (n, 3, h * self.scale, w * self.scale)
        out = self._check_out(out, shape, torch.float32)
        if n == 0:      # an empty batch gives an empty result, as the reference modules do (nothing to launch)
            return out
        check(lib().srmi_model_forward_f32(self._h, x.data_ptr(), out.data_ptr(), n, h, w, self._stream()),
              "srmi_model_forward_f32")
        return out

    def forward_u8(self, img: torch.Tensor, round_mode=_lib.ROUND_HALF_UP, out: torch.Tensor = None) -> torch.Tensor:
        """img: uint8 NHWC (or HWC) on this engine's GPU -> uint8 NHWC, runner pre/post fused."""
        squeeze = img.dim() == 3
        if squeeze:
            img = img.unsqueeze(0)
        if img.dim() != 4 or img.shape[3] != 3 or img.dtype != torch.uint8:
            raise ValueError(f"expected uint8 [N,H,W,3], got {img.dtype} {tuple(img.shape)}")
        img = img.to(self.device).contiguous()
        if out is None:
            y = torch.ops.srmi355.forward_u8(self.handle(), self.scale, img, int(round_mode))
            return y[0] if squeeze else y
        n, h, w, _ = img.shape
        shape = (n, h * self.scale, w * self.scale, 3)
        out = self._check_out(out, shape, torch.uint8)
        if n == 0:
            return out
        check(lib().srmi_model_forward_u8(self._h, img.data_ptr(), out.data_ptr(), n, h, w, round_mode,
                                          self._stream()), "srmi_model_forward_u8")
        return out[0] if squeeze else out


class HipModule(nn.Module):
    """nn.Module whose parameters live in the reference's module tree (so the official
    checkpoints load with the reference's own key names) and whose forward runs on the GPU
    through an `Engine`.  Parameters are re-uploaded automatically when they change."""

    def __init__(self, compute_dtype=None, trunk_f32=True, device=None):
        super().__init__()
        self._srmi_dtype = resolve_dtype(compute_dtype)
        self._srmi_trunk_f32 = bool(trunk_f32)
        self._srmi_device = device
        self._srmi_engine = None
        self._srmi_stamp = None
        self._srmi_handle = None   # int handle and scale of the current engine, for the traced (torch.compile) path
        self._srmi_scale = None

    # subclasses return the Engine constructor kwargs
    def _engine_config(self):
        raise NotImplementedError

    def _param_stamp(self):
        return tuple((p._version, p.data_ptr()) for p in self.parameters())

    def engine(self) -> Engine:
        """The up-to-date executor (uploads the weights on first use and after any change)."""
        stamp = self._param_stamp()
        if self._srmi_engine is None or stamp != self._srmi_stamp:
            dev = self._srmi_device
            if dev is None:
                first = next(iter(self.parameters()))
                dev = first.device.index if first.is_cuda else 0
            if self._srmi_engine is None:
                self._srmi_engine = Engine(dtype=self._srmi_dtype, trunk_f32=self._srmi_trunk_f32, device=dev or 0,
                                           **self._engine_config())
            sd = self.state_dict()
            want = self._srmi_engine.keys()
            self._srmi_engine.load_state_dict({k: sd[k] for k in want})
            self._srmi_stamp = stamp
            self._srmi_handle, self._srmi_scale = self._srmi_engine.handle(), int(self._srmi_engine.scale)
        return self._srmi_engine

    def forward(self, x):
        """f32 NCHW in -> f32 NCHW out, the reference modules' forward contract.  The result is
        returned on the input's device.

        Under torch.compile the call is one node, torch.ops.srmi355.forward_f32 on the engine built by the last eager
        `model.engine()` / `model(x)` call (uploading weights is host work that cannot be traced): build the engine
        once before compiling, and again after changing parameters."""
        if torch.compiler.is_compiling():
            if self._srmi_handle is None:
                raise RuntimeError("call model.engine() (or run one eager forward) before torch.compile: the weights are "
                                   "uploaded outside the traced region")
            return torch.ops.srmi355.forward_f32(self._srmi_handle, self._srmi_scale, x)
        eng = self.engine()
        y = eng.forward_f32(x)
        return y if x.is_cuda else y.to(x.device)

    def forward_u8(self, img, round_mode=_lib.ROUND_HALF_UP):
        """u8 HWC/NHWC in -> u8 out with the runners' pre/post-processing fused."""
        eng = self.engine()
        t = torch.as_tensor(img)
        y = eng.forward_u8(t, round_mode)
        return y if t.is_cuda else y.cpu()


def conv_params(cin, cout, k):
    """A parameter container with nn.Conv2d's names, shapes and default initialisation."""
    return nn.Conv2d(cin, cout, k, padding=k // 2, bias=True)
