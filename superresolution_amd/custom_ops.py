"""The forward passes as registered PyTorch operators (`torch.ops.srmi355.forward_f32`, `forward_u8`), over the same C
ABI the ctypes surface uses (srmi_model_forward_f32 / srmi_model_forward_u8 in include/srmi355.h).

The reference calls `model(img)` under `torch.no_grad()` (realesrgan/mtk/python/test/test_pytorch.py:127-128); registered
as operators the same call is visible to the dispatcher, carries a shape function for fake tensors, and can sit inside
`torch.compile(fullgraph=True)`: the whole network is ONE node of the captured graph.  The engine is named by its handle
(an integer), which the caller keeps alive (HipModule does).  Inference only: no autograd formula is registered.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import check, lib


def _stream(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


@torch.library.custom_op("srmi355::forward_f32", mutates_args=(), device_types="cuda")
def forward_f32(handle: int, scale: int, x: torch.Tensor) -> torch.Tensor:
    """x: float32 NCHW [n,3,h,w] on the engine's GPU -> float32 NCHW [n,3,h*scale,w*scale]."""
    if x.dim() != 4 or x.shape[1] != 3:
        raise ValueError(f"expected [N,3,H,W], got {tuple(x.shape)}")
    x = x.to(torch.float32).contiguous()
    n, _, h, w = x.shape
    out = torch.empty((n, 3, h * scale, w * scale), dtype=torch.float32, device=x.device)
    if n:
        check(lib().srmi_model_forward_f32(C.c_void_p(handle), x.data_ptr(), out.data_ptr(), n, h, w, _stream(x.device)),
              "srmi_model_forward_f32")
    return out


@forward_f32.register_fake
def _(handle, scale, x):
    torch._check(x.dim() == 4)
    n, _, h, w = x.shape
    return x.new_empty((n, 3, h * scale, w * scale), dtype=torch.float32)


@torch.library.custom_op("srmi355::forward_u8", mutates_args=(), device_types="cuda")
def forward_u8(handle: int, scale: int, img: torch.Tensor, round_mode: int) -> torch.Tensor:
    """img: uint8 NHWC [n,h,w,3] on the engine's GPU -> uint8 NHWC [n,h*scale,w*scale,3], the runners' pre/post fused."""
    if img.dim() != 4 or img.shape[3] != 3 or img.dtype != torch.uint8:
        raise ValueError(f"expected uint8 [N,H,W,3], got {img.dtype} {tuple(img.shape)}")
    img = img.contiguous()
    n, h, w, _ = img.shape
    out = torch.empty((n, h * scale, w * scale, 3), dtype=torch.uint8, device=img.device)
    if n:
        check(lib().srmi_model_forward_u8(C.c_void_p(handle), img.data_ptr(), out.data_ptr(), n, h, w, round_mode,
                                          _stream(img.device)), "srmi_model_forward_u8")
    return out


@forward_u8.register_fake
def _(handle, scale, img, round_mode):
    torch._check(img.dim() == 4)
    n, h, w, _ = img.shape
    return img.new_empty((n, h * scale, w * scale, 3), dtype=torch.uint8)
