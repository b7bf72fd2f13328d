"""EDSR on MI355X — same construction, state_dict layout and forward(x) contract as the reference
module (/root/reference/edsr/mtk/python/edsr_model.py), executed by HIP kernels.

  default_conv                                   edsr_model.py:11
  MeanShift(rgb_range, rgb_mean, rgb_std, sign)  edsr_model.py:18
  EDSR(n_resblocks=32, n_feats=256, scale=4, rgb_range=255, res_scale=0.1, conv=default_conv)  :82
  EDSR.load_state_dict (name-matched copy, tolerant only for 'tail' keys)   :126
  EDSRCore (no MeanShift)                        :146
  load_edsr_from_checkpoint(path, scale=4)       :183
  create_core_model_from_full / get_meanshift_params   :199 / :219
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import _lib
from .engine import HipModule, conv_params

RGB_MEAN = (0.4488, 0.4371, 0.4040)


def default_conv(in_channels, out_channels, kernel_size, bias=True):
    return nn.Conv2d(in_channels, out_channels, kernel_size, padding=kernel_size // 2, bias=bias)


class MeanShift(nn.Conv2d):
    """Frozen 1x1 conv: weight = I/std, bias = sign*rgb_range*mean/std (edsr_model.py:18-26).
    Its tensors are part of the state_dict; the engine applies them as a 3x3 colour matrix in the
    first conv's loader (in-bounds pixels only) and in the last conv's epilogue."""

    def __init__(self, rgb_range, rgb_mean=RGB_MEAN, rgb_std=(1.0, 1.0, 1.0), sign=-1):
        super().__init__(3, 3, kernel_size=1)
        std = torch.tensor(rgb_std, dtype=torch.float32)
        self.weight.data = torch.eye(3).view(3, 3, 1, 1) / std.view(3, 1, 1, 1)
        self.bias.data = sign * rgb_range * torch.tensor(rgb_mean, dtype=torch.float32) / std
        for p in self.parameters():
            p.requires_grad = False


class ResBlock(nn.Module):
    """Parameter container: body.0 (conv, then ReLU) and body.2 (conv), indices as in the
    reference's nn.Sequential(conv, act, conv) (edsr_model.py:31-42)."""

    def __init__(self, conv, n_feats, kernel_size, bias=True, bn=False, act=None, res_scale=1):
        # the reference's signature (edsr_model.py:31); the engine implements what the reference networks use
        super().__init__()
        _check_block_options(conv, kernel_size, bias, bn, act)
        self.body = nn.Sequential(conv_params(n_feats, n_feats, 3), nn.ReLU(True), conv_params(n_feats, n_feats, 3))
        self.res_scale = res_scale


class Upsampler(nn.Sequential):
    """conv C->4C + PixelShuffle(2) per factor of two, or conv C->9C + PixelShuffle(3)
    (edsr_model.py:50-77); conv indices 0, 2, ... as in the reference."""

    def __init__(self, conv, scale, n_feats, bn=False, act=False, bias=True):
        # the reference's signature (edsr_model.py:52)
        _check_block_options(conv, 3, bias, bn, None if act in (False, None) else act)
        m = []
        if scale >= 2 and (scale & (scale - 1)) == 0:
            for _ in range(int(round(math.log2(scale)))):
                m += [conv_params(n_feats, 4 * n_feats, 3), nn.PixelShuffle(2)]
        elif scale == 3:
            m += [conv_params(n_feats, 9 * n_feats, 3), nn.PixelShuffle(3)]
        else:
            raise NotImplementedError(f"scale {scale} is not supported")
        super().__init__(*m)


def _check_block_options(conv, kernel_size, bias, bn, act):
    """Options of the reference's building blocks that its three networks never use have no kernel here."""
    _check_conv(conv)
    if kernel_size != 3 or not bias or bn:
        raise NotImplementedError("the HIP engine implements 3x3 convolutions with bias and without batch norm")
    if act is not None and not isinstance(act, nn.ReLU):
        raise NotImplementedError("the HIP engine implements the ReLU blocks of the reference networks")


def _check_conv(conv):
    if conv is not default_conv and conv is not None:
        raise NotImplementedError("the HIP engine implements the default 3x3 convolution only")


class _EDSRBase(HipModule):
    def __init__(self, n_resblocks, n_feats, scale, res_scale, with_meanshift, rgb_range, engine_kwargs):
        super().__init__(**engine_kwargs)
        self.n_resblocks, self.n_feats, self.scale, self.res_scale = n_resblocks, n_feats, scale, res_scale
        self._with_meanshift = with_meanshift
        if with_meanshift:
            self.sub_mean = MeanShift(rgb_range)
            self.add_mean = MeanShift(rgb_range, sign=1)
        self.head = nn.Sequential(conv_params(3, n_feats, 3))
        self.body = nn.Sequential(*[ResBlock(default_conv, n_feats, 3, act=nn.ReLU(True), res_scale=res_scale)
                                    for _ in range(n_resblocks)],
                                  conv_params(n_feats, n_feats, 3))
        self.tail = nn.Sequential(Upsampler(default_conv, scale, n_feats, act=False), conv_params(n_feats, 3, 3))

    def _engine_config(self):
        return dict(arch=_lib.ARCH_EDSR, scale=self.scale, n_feats=self.n_feats, n_blocks=self.n_resblocks,
                    res_scale=self.res_scale, with_meanshift=self._with_meanshift)


class EDSR(_EDSRBase):
    """x: f32 NCHW in 0..rgb_range -> add_mean(tail(body(head(sub_mean(x))) + head))
    (edsr_model.py:114-124)."""

    def __init__(self, n_resblocks=32, n_feats=256, scale=4, rgb_range=255, res_scale=0.1, conv=default_conv,
                 *, compute_dtype=None, trunk_f32=True, device=None):
        _check_conv(conv)
        super().__init__(n_resblocks, n_feats, scale, res_scale, True, rgb_range,
                         dict(compute_dtype=compute_dtype, trunk_f32=trunk_f32, device=device))

    def load_state_dict(self, state_dict, strict=True):
        """The reference's own loader semantics (edsr_model.py:126-143): copy by name; a shape
        mismatch raises unless the key belongs to the tail; unexpected keys raise under strict
        unless they belong to the tail; missing keys are not an error."""
        own = self.state_dict()
        for name, param in state_dict.items():
            if name in own:
                try:
                    own[name].copy_(param.data if isinstance(param, nn.Parameter) else param)
                except Exception:
                    if "tail" not in name:
                        raise RuntimeError(
                            f"While copying the parameter named {name}, whose dimensions in the model are "
                            f"{own[name].size()} and whose dimensions in the checkpoint are {param.size()}.")
            elif strict and "tail" not in name:
                raise KeyError(f'unexpected key "{name}" in state_dict')


class EDSRCore(_EDSRBase):
    """EDSR without the MeanShift layers (edsr_model.py:146-180): the caller subtracts / adds
    255*mean, as the runners' pre/post-processing does."""

    def __init__(self, n_resblocks=32, n_feats=256, scale=4, res_scale=0.1, conv=default_conv,
                 *, compute_dtype=None, trunk_f32=True, device=None):
        _check_conv(conv)
        super().__init__(n_resblocks, n_feats, scale, res_scale, False, 255,
                         dict(compute_dtype=compute_dtype, trunk_f32=trunk_f32, device=device))


def load_edsr_from_checkpoint(checkpoint_path, scale=4, **engine_kwargs):
    """EDSR(scale) with weights from a bare state_dict or one under 'model' (edsr_model.py:183-196)."""
    ckpt = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
    model = EDSR(scale=scale, **engine_kwargs)
    model.load_state_dict(ckpt["model"] if "model" in ckpt else ckpt, strict=False)
    return model


def create_core_model_from_full(full_model, scale=4):
    """EDSRCore carrying the full model's weights minus MeanShift (edsr_model.py:199-216)."""
    core = EDSRCore(n_resblocks=full_model.n_resblocks, n_feats=full_model.n_feats, scale=scale,
                    res_scale=full_model.res_scale,
                    compute_dtype=full_model._srmi_dtype, trunk_f32=full_model._srmi_trunk_f32,
                    device=full_model._srmi_device)
    full = full_model.state_dict()
    sd = core.state_dict()
    copied = 0
    for k in sd:
        if k in full:
            sd[k].copy_(full[k])
            copied += 1
    print(f"  copied {copied}/{len(sd)} tensors")
    return core


def get_meanshift_params():
    return {"rgb_mean": [0.4488, 0.4371, 0.4040], "rgb_range": 255.0}
