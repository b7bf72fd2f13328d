"""Operator-level Python surface over the C ABI (one HIP kernel launch per call).

`Planes` wraps a device buffer in the library's plane layout [N][C/CPB][H][W][CPB]
(CPB = 16 for f16/bf16, 8 for f32: 32 bytes per pixel per plane).  The functions mirror the torch ops the reference modules
call; they exist for tests, interop and experimentation — whole networks run through
`srmi_model_*` (see engine.py) so that a forward pass is a single ctypes call.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import F32, F16, BF16, I8, Tensor, check, lib

_TORCH_DT = {F32: torch.float32, F16: torch.float16, BF16: torch.bfloat16, I8: torch.int8}


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


PIX_BYTES = 32  # bytes of channels per pixel in one plane


def cpb(dtype):
    return 8 if dtype == F32 else 32 if dtype == I8 else 16


class Planes:
    """A plane-layout activation tensor on the GPU."""

    def __init__(self, n, c, h, w, dtype, device="cuda", zero=False, qscale=0.0):
        self.n, self.c, self.h, self.w, self.dtype = n, c, h, w, dtype
        self.qscale = float(qscale)   # int8 tensors: stored q = clamp(rint(value * qscale), -127, 127)
        self.planes = (c + cpb(dtype) - 1) // cpb(dtype)
        shape = (n, self.planes, h, w, cpb(dtype))
        self.buf = (torch.zeros if zero else torch.empty)(shape, dtype=_TORCH_DT[dtype], device=device)

    def view(self, c0=0):
        """C struct for the sub-tensor starting at channel c0 (a multiple of the plane width)."""
        assert c0 % cpb(self.dtype) == 0
        t = Tensor()
        es = self.buf.element_size()
        t.ptr = self.buf.data_ptr() + (c0 // cpb(self.dtype)) * self.h * self.w * PIX_BYTES
        t.img_stride = self.planes * self.h * self.w * PIX_BYTES // es
        t.plane_stride = self.h * self.w * PIX_BYTES // es
        t.dtype = self.dtype
        t.qscale = self.qscale
        return t

    @staticmethod
    def from_nchw(x: torch.Tensor, dtype, qscale=0.0):
        x = x.to(device="cuda", dtype=torch.float32).contiguous()
        n, c, h, w = x.shape
        p = Planes(n, c, h, w, dtype, qscale=qscale)
        v = p.view()
        check(lib().srmi_nchw_to_planes(x.data_ptr(), C.byref(v), n, c, h, w, _stream()), "nchw_to_planes")
        return p

    def to_nchw(self) -> torch.Tensor:
        out = torch.empty((self.n, self.c, self.h, self.w), dtype=torch.float32, device=self.buf.device)
        v = self.view()
        check(lib().srmi_planes_to_nchw(C.byref(v), out.data_ptr(), self.n, self.c, self.h, self.w, _stream()),
              "planes_to_nchw")
        return out


_NULL = Tensor()


class PackedConv:
    """Weights of one 3x3 convolution in the kernel's device format."""

    def __init__(self, weight, bias, dtype, cout_tile=None, ps_r=1, device="cuda", wscale=None, winograd=False,
                 subpixel_up2=False):
        """dtype I8: `weight` holds integer values in [-127, 127] and `wscale` the per-output-channel factor the
        int32 accumulator is multiplied with (weight scale * input activation scale).
        winograd: pack for the Winograd F(2x2,3x3) form of the convolution (f16, cout_tile 32).
        subpixel_up2: pack the sub-pixel form of a convolution that is used with up2=True (f16 / bf16)."""
        w = np.ascontiguousarray(np.asarray(weight, dtype=np.float32))
        cout, cin = w.shape[:2]
        assert w.shape[2:] == (3, 3)
        if cout_tile is None:
            cout_tile = 32 if (cout <= 32 or winograd) else 64
        self.cout, self.cin, self.dtype, self.cout_tile, self.ps_r = cout, cin, dtype, cout_tile, ps_r
        # winograd: False / 0 (direct), True / 1 (F(2x2,3x3)), 2 (F(2,3) along x)
        self.algo = _lib.ALGO_SUBPIXEL_UP2 if subpixel_up2 else int(winograd)
        nb = C.c_int(0)
        b = None if bias is None else np.ascontiguousarray(np.asarray(bias, dtype=np.float32))
        if subpixel_up2:
            if winograd or ps_r != 1:
                raise _lib.SrmiError("the sub-pixel form excludes the Winograd forms and the pixel-shuffle store")
            nbytes = lib().srmi_pack_conv3x3_up2_bytes(dtype, cout, cin, cout_tile, C.byref(nb))
        elif winograd:
            if dtype != F16 or cout_tile != 32 or ps_r != 1:
                raise _lib.SrmiError("the Winograd form needs f16, cout_tile 32 and no pixel shuffle")
            nbytes = (lib().srmi_pack_wino1d3x3_bytes if self.algo == _lib.ALGO_WINOGRAD_X else
                      lib().srmi_pack_wino3x3_bytes)(cout, cin, C.byref(nb))
        else:
            nbytes = lib().srmi_pack_conv3x3_bytes(dtype, cout, cin, cout_tile, C.byref(nb))
        if nbytes == 0:
            raise _lib.SrmiError(f"cannot pack conv weight {w.shape} for dtype {dtype}")
        wp = np.empty(nbytes, dtype=np.uint8)
        bp = np.empty(nb.value, dtype=np.float32)
        if subpixel_up2:
            check(lib().srmi_pack_conv3x3_up2(dtype, cout, cin, cout_tile, w.ctypes.data, None if b is None else b.ctypes.data,
                                              wp.ctypes.data, bp.ctypes.data), "pack_conv3x3_up2")
        elif winograd:
            pack = lib().srmi_pack_wino1d3x3 if self.algo == _lib.ALGO_WINOGRAD_X else lib().srmi_pack_wino3x3
            check(pack(cout, cin, w.ctypes.data, None if b is None else b.ctypes.data, wp.ctypes.data, bp.ctypes.data),
                  "pack_wino3x3")
        else:
            check(lib().srmi_pack_conv3x3(dtype, cout, cin, cout_tile, ps_r, w.ctypes.data,
                                          None if b is None else b.ctypes.data, wp.ctypes.data, bp.ctypes.data),
                  "pack_conv3x3")
        self.host_w, self.host_b = wp, bp
        self.s = None
        if device is not None:
            self.w = torch.from_numpy(wp).to(device)
            self.b = torch.from_numpy(bp).to(device)
            if wscale is not None:
                sp = np.zeros(nb.value, dtype=np.float32)
                sp[:cout] = np.asarray(wscale, dtype=np.float32)
                self.s = torch.from_numpy(sp).to(device)


def conv3x3(x: Planes, pc: PackedConv, *, act=0, up2=False, alpha=1.0, res1: Planes = None,
            beta=1.0, res2: Planes = None, out: Planes = None, out_c0=0, out_dtype=None,
            out2: Planes = None, chansum=False, rows_per_wave=0, tail=None, out_round8=False, out_qscale=0.0,
            in_gate: Planes = None, gate: torch.Tensor = None, gate_alpha=1.0, in_sum: Planes = None, attention=None):
    """act(conv(x)+b)*alpha+res1 (*beta+res2) -> out (see srmi_conv3x3 in include/srmi355.h).

    in_gate / gate / gate_alpha: the conv reads x + gate_alpha * (in_gate * gate[n][c]) (gate: f32 [n][cin] on the
    GPU) and stores that sum to in_sum (if given).
    attention: (w1 [cr][c], b1, w2 [c][cr], b2) of a CALayer; the conv then also returns its scale s[n][c]
    (pooling, squeeze / excite and sigmoid finished by the last workgroup of each image).

    tail: None, or dict(mode=TAIL_*, mat=3x3, bias=3, lo, hi, scale, round) for the RGB layer.
    Returns out (Planes), or the tail tensor; with chansum=True returns (out, sums[n][tiles][c]).
    """
    n = x.n
    h, w = (2 * x.h, 2 * x.w) if up2 else (x.h, x.w)
    d = _lib.ConvDesc()
    d.dtype = pc.dtype
    d.n, d.h, d.w, d.cin, d.cout = n, h, w, pc.cin, pc.cout
    d.in_up2, d.act, d.ps_r, d.cout_tile, d.rows_per_wave = int(up2), act, pc.ps_r, pc.cout_tile, rows_per_wave
    d.algo = getattr(pc, "algo", _lib.ALGO_DIRECT)
    d.inp = x.view()
    d.wpk, d.bias = pc.w.data_ptr(), pc.b.data_ptr()
    d.wscale = pc.s.data_ptr() if pc.s is not None else None
    d.alpha, d.beta = alpha, beta
    d.res1 = res1.view() if res1 is not None else _NULL
    d.res2 = res2.view() if res2 is not None else _NULL
    keep = [x, pc, res1, res2, in_gate, gate, in_sum]
    if in_gate is not None:
        d.in_gate, d.gate, d.gate_alpha = in_gate.view(), gate.data_ptr(), gate_alpha
        d.in_sum = in_sum.view() if in_sum is not None else _NULL
    sums = None
    result = None
    att_scale = None
    if attention is not None:
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(np.asarray(a, np.float32).reshape(-1))).cuda()
        w1, b1, w2, b2 = (dev(a) for a in attention)
        att_sums = torch.zeros((n, _lib.CA_SLOTS, pc.cout), dtype=torch.int64, device="cuda")
        att_count = torch.zeros((n,), dtype=torch.int32, device="cuda")
        att_scale = torch.empty((n, pc.cout), dtype=torch.float32, device="cuda")
        d.ca_sums, d.ca_count, d.ca_scale = att_sums.data_ptr(), att_count.data_ptr(), att_scale.data_ptr()
        d.ca_w1, d.ca_b1, d.ca_w2, d.ca_b2 = w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr()
        d.ca_cr = b1.numel()
        keep += [w1, b1, w2, b2, att_sums, att_count]
    if tail is None:
        r = pc.ps_r
        if out is None:
            out = Planes(n, pc.cout // (r * r), h * r, w * r, pc.dtype if out_dtype is None else out_dtype, qscale=out_qscale)
        d.out = out.view(out_c0)
        if out_round8:
            d.out.flags = _lib.TENSOR_ROUND8
        d.out2 = out2.view() if out2 is not None else _NULL
        d.tail_mode = 0
        if chansum:
            tiles = lib().srmi_conv3x3_tiles(h, w, rows_per_wave)
            sums = torch.empty((n, tiles, pc.cout), dtype=torch.float32, device="cuda")
            d.chansum = sums.data_ptr()
        result = out
    else:
        d.tail_mode = tail["mode"]
        if tail["mode"] == _lib.TAIL_F32_NCHW:
            result = torch.empty((n, 3, h, w), dtype=torch.float32, device="cuda")
        else:
            result = torch.empty((n, h, w, 3), dtype=torch.uint8, device="cuda")
        d.tail_out = result.data_ptr()
        mat = np.asarray(tail.get("mat", np.eye(3)), dtype=np.float32).reshape(9)
        tb = np.asarray(tail.get("bias", np.zeros(3)), dtype=np.float32).reshape(3)
        for i in range(9):
            d.tail_mat[i] = float(mat[i])
        for i in range(3):
            d.tail_bias[i] = float(tb[i])
        d.tail_lo, d.tail_hi = tail.get("lo", 0.0), tail.get("hi", 1.0)
        d.tail_scale, d.tail_round = tail.get("scale", 255.0), tail.get("round", 0)
    check(lib().srmi_conv3x3(C.byref(d), _stream()), "conv3x3")
    if attention is not None:
        torch.cuda.current_stream().synchronize()
        if int(att_sums.abs().max()) or int(att_count.abs().max()):
            raise _lib.SrmiError("conv3x3: attention accumulators were not left at zero")
        del keep
        return result, att_scale
    del keep
    if chansum:
        return result, sums
    return result


def rcab_fused(x: Planes, pc1: PackedConv, pc2: PackedConv, *, in_gate: Planes = None, gate: torch.Tensor = None,
               gate_alpha=1.0, in_sum: Planes = None, attention=None, out: Planes = None):
    """conv2(relu(conv1(x'))) in one launch, x' = x + gate_alpha * (in_gate * gate[n][c]) (srmi_rcab_fused).  Without
    in_gate the block's input is x itself.  attention: (w1, b1, w2, b2) of the CALayer -> also returns scale[n][c]."""
    n, c, h, w = x.n, x.c, x.h, x.w
    d = _lib.RcabDesc()
    # compute type: the body tensor's (x may be the f32 residual trunk when the input is gated)
    cdt = in_gate.dtype if in_gate is not None else x.dtype
    d.dtype, d.n, d.h, d.w, d.c = cdt, n, h, w, c
    d.inp = x.view()
    keep = [x, pc1, pc2, in_gate, gate, in_sum]
    if in_gate is not None:
        d.in_gate, d.gate, d.gate_alpha = in_gate.view(), gate.data_ptr(), gate_alpha
        d.in_sum = in_sum.view() if in_sum is not None else _NULL
    else:
        zeros = torch.zeros((n, c), dtype=torch.float32, device="cuda")
        keep.append(zeros)
        d.in_gate, d.gate, d.gate_alpha = x.view(), zeros.data_ptr(), 0.0
    d.w1, d.b1, d.w2, d.b2 = pc1.w.data_ptr(), pc1.b.data_ptr(), pc2.w.data_ptr(), pc2.b.data_ptr()
    if out is None:
        out = Planes(n, c, h, w, cdt)
    d.out = out.view()
    att_scale = None
    if attention is not None:
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(np.asarray(a, np.float32).reshape(-1))).cuda()
        w1, b1, w2, b2 = (dev(a) for a in attention)
        att_sums = torch.zeros((n, _lib.CA_SLOTS, c), dtype=torch.int64, device="cuda")
        att_count = torch.zeros((n,), dtype=torch.int32, device="cuda")
        att_scale = torch.empty((n, c), dtype=torch.float32, device="cuda")
        d.ca_sums, d.ca_count, d.ca_scale = att_sums.data_ptr(), att_count.data_ptr(), att_scale.data_ptr()
        d.ca_w1, d.ca_b1, d.ca_w2, d.ca_b2 = w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr()
        d.ca_cr = b1.numel()
        keep += [w1, b1, w2, b2, att_sums, att_count]
    check(lib().srmi_rcab_fused(C.byref(d), _stream()), "rcab_fused")
    if attention is not None:
        torch.cuda.current_stream().synchronize()
        if int(att_sums.abs().max()) or int(att_count.abs().max()):
            raise _lib.SrmiError("rcab_fused: attention accumulators were not left at zero")
        del keep
        return out, att_scale
    del keep
    return out


def conv_first(inp: torch.Tensor, weight, bias, dtype, *, pre_mat=None, pre_bias=None, out2_dtype=None):
    """First (3-channel) layer; inp is f32 NCHW or u8 NHWC on the GPU."""
    w = np.ascontiguousarray(np.asarray(weight, dtype=np.float32))  # [cout][3][3][3] OIHW
    cout = w.shape[0]
    wk = torch.from_numpy(np.ascontiguousarray(w.transpose(2, 3, 1, 0))).cuda()  # ky kx cin cout
    b = torch.from_numpy(np.ascontiguousarray(np.asarray(bias, dtype=np.float32))).cuda()
    d = _lib.ConvFirstDesc()
    if inp.dtype == torch.uint8:
        n, h, ww, _ = inp.shape
        d.in_kind = _lib.IN_U8_NHWC
    else:
        n, _, h, ww = inp.shape
        d.in_kind = _lib.IN_F32_NCHW
    inp = inp.contiguous()
    d.n, d.h, d.w, d.cout = n, h, ww, cout
    d.inp = inp.data_ptr()
    pm = np.eye(3, dtype=np.float32) if pre_mat is None else np.asarray(pre_mat, np.float32)
    pb = np.zeros(3, np.float32) if pre_bias is None else np.asarray(pre_bias, np.float32)
    for i in range(9):
        d.pre_mat[i] = float(pm.reshape(9)[i])
    for i in range(3):
        d.pre_bias[i] = float(pb[i])
    d.w_kcio, d.bias = wk.data_ptr(), b.data_ptr()
    out = Planes(n, cout, h, ww, dtype)
    d.out = out.view()
    out2 = None
    if out2_dtype is not None:
        out2 = Planes(n, cout, h, ww, out2_dtype)
        d.out2 = out2.view()
    check(lib().srmi_conv_first(C.byref(d), _stream()), "conv_first")
    torch.cuda.current_stream().synchronize()  # wk/b are temporaries
    return (out, out2) if out2 is not None else out


def ca_scale(sums: torch.Tensor, hw, w1, b1, w2, b2):
    n, tiles, c = sums.shape
    cr = w1.shape[0]
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(np.asarray(a, np.float32).reshape(-1))).cuda()
    W1, B1, W2, B2 = dev(w1), dev(b1), dev(w2), dev(b2)
    scale = torch.empty((n, c), dtype=torch.float32, device="cuda")
    check(lib().srmi_ca_scale(sums.data_ptr(), n, tiles, c, cr, hw, W1.data_ptr(), B1.data_ptr(),
                              W2.data_ptr(), B2.data_ptr(), scale.data_ptr(), _stream()), "ca_scale")
    torch.cuda.current_stream().synchronize()
    return scale


def ca_apply(y: Planes, x: Planes, scale: torch.Tensor, res_scale=1.0, out_dtype=None, out2_dtype=None):
    out = Planes(y.n, y.c, y.h, y.w, y.dtype if out_dtype is None else out_dtype)
    out2 = Planes(y.n, y.c, y.h, y.w, out2_dtype) if out2_dtype is not None else None
    yv, xv, ov = y.view(), x.view(), out.view()
    o2 = out2.view() if out2 is not None else None
    check(lib().srmi_ca_apply(C.byref(yv), C.byref(xv), scale.data_ptr(), res_scale, y.n, y.h, y.w, y.c,
                              C.byref(ov), C.byref(o2) if o2 is not None else None, _stream()), "ca_apply")
    return (out, out2) if out2 is not None else out


def channel_sums(x: Planes):
    sums = torch.empty((x.n, x.c), dtype=torch.float32, device="cuda")
    v = x.view()
    check(lib().srmi_channel_sums(C.byref(v), x.n, x.h, x.w, x.c, sums.data_ptr(), _stream()), "channel_sums")
    return sums


def pixel_shuffle(x: Planes, r):
    out = Planes(x.n, x.c // (r * r), x.h * r, x.w * r, x.dtype)
    a, b = x.view(), out.view()
    check(lib().srmi_pixel_shuffle(C.byref(a), C.byref(b), x.n, x.h, x.w, x.c, r, _stream()), "pixel_shuffle")
    return out


def nearest_up2(x: Planes):
    out = Planes(x.n, x.c, 2 * x.h, 2 * x.w, x.dtype)
    a, b = x.view(), out.view()
    check(lib().srmi_nearest_up2(C.byref(a), C.byref(b), x.n, x.h, x.w, x.c, _stream()), "nearest_up2")
    return out


def resize_bilinear_u8(img: torch.Tensor, out_w, out_h):
    """img: u8 HWC on the GPU."""
    img = img.contiguous()
    in_h, in_w, _ = img.shape
    out = torch.empty((out_h, out_w, 3), dtype=torch.uint8, device=img.device)
    check(lib().srmi_resize_bilinear_u8(img.data_ptr(), in_w, in_h, out.data_ptr(), out_w, out_h, _stream()),
          "resize_bilinear_u8")
    return out
