"""RCAN on MI355X — same construction, state_dict layout and forward(x) contract as the reference
module (/root/reference/rcan/mtk/python/rcan_model.py), executed by HIP kernels.

  RCAN(scale, n_resgroups=10, n_resblocks=20, n_feats=64, reduction=16, res_scale=1.0,
       rgb_range=255.0)                                  rcan_model.py:123
  RCANCore (no MeanShift)                                rcan_model.py:166
  CALayer / RCAB / ResidualGroup parameter trees         rcan_model.py:61-112
  load_rcan_from_checkpoint(path, scale)                 rcan_model.py:205
  create_core_model_from_full / get_meanshift_params / infer_scale_from_model_name   :230 / :256 / :268
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib
from .edsr_model import MeanShift as _IdentityMeanShift
from .edsr_model import Upsampler, _check_block_options, default_conv  # noqa: F401  (same layers, rcan_model.py:17-58)
from .engine import HipModule, conv_params


class MeanShift(_IdentityMeanShift):
    """RCAN's MeanShift.  The reference initialiser only divides the randomly initialised 1x1
    weight by std (rcan_model.py:28-34) and relies on the checkpoint to overwrite it with the
    identity; here it starts as the identity, which is what every official checkpoint holds and
    what the reference's second copy does (rcan/rknn/python/test_pytorch.py:22-32)."""


class CALayer(nn.Module):
    """Parameter container of channel attention: conv_du.0 (C -> C/r, ReLU) and conv_du.2
    (C/r -> C, Sigmoid) (rcan_model.py:63-71)."""

    def __init__(self, channel, reduction=16):
        super().__init__()
        self.avg_pool = nn.AdaptiveAvgPool2d(1)
        self.conv_du = nn.Sequential(nn.Conv2d(channel, channel // reduction, 1, padding=0, bias=True), nn.ReLU(True),
                                     nn.Conv2d(channel // reduction, channel, 1, padding=0, bias=True), nn.Sigmoid())


class RCAB(nn.Module):
    """body.0 conv (+ReLU), body.2 conv, body.3 CALayer (rcan_model.py:81-92)."""

    def __init__(self, conv, n_feats, kernel_size, reduction, bias=True, bn=False, act=None, res_scale=1):
        # the reference's signature (rcan_model.py:81)
        super().__init__()
        _check_block_options(conv, kernel_size, bias, bn, act)
        self.body = nn.Sequential(conv_params(n_feats, n_feats, 3), nn.ReLU(True), conv_params(n_feats, n_feats, 3),
                                  CALayer(n_feats, reduction))
        self.res_scale = res_scale


class ResidualGroup(nn.Module):
    """body.0..n-1 RCABs, body.n conv (rcan_model.py:102-107)."""

    def __init__(self, conv, n_feats, kernel_size, reduction, act, res_scale, n_resblocks):
        # the reference's signature (rcan_model.py:102)
        super().__init__()
        _check_block_options(conv, kernel_size, True, False, act)
        self.body = nn.Sequential(*[RCAB(conv, n_feats, kernel_size, reduction, act=act, res_scale=res_scale)
                                    for _ in range(n_resblocks)],
                                  conv_params(n_feats, n_feats, 3))


class _RCANBase(HipModule):
    def __init__(self, scale, n_resgroups, n_resblocks, n_feats, reduction, res_scale, with_meanshift, rgb_range,
                 engine_kwargs):
        super().__init__(**engine_kwargs)
        self.scale, self.n_resgroups, self.n_resblocks = scale, n_resgroups, n_resblocks
        self.n_feats, self.reduction, self.res_scale = n_feats, reduction, res_scale
        self._with_meanshift = with_meanshift
        if with_meanshift:
            self.sub_mean = MeanShift(rgb_range)
            self.add_mean = MeanShift(rgb_range, sign=1)
        self.head = nn.Sequential(conv_params(3, n_feats, 3))
        self.body = nn.Sequential(*[ResidualGroup(default_conv, n_feats, 3, reduction, nn.ReLU(True), res_scale, n_resblocks)
                                    for _ in range(n_resgroups)], conv_params(n_feats, n_feats, 3))
        self.tail = nn.Sequential(Upsampler(default_conv, scale, n_feats, act=False), conv_params(n_feats, 3, 3))

    def _engine_config(self):
        return dict(arch=_lib.ARCH_RCAN, scale=self.scale, n_feats=self.n_feats, n_blocks=self.n_resblocks,
                    n_groups=self.n_resgroups, reduction=self.reduction, res_scale=self.res_scale,
                    with_meanshift=self._with_meanshift)


class RCAN(_RCANBase):
    """x: f32 NCHW in 0..rgb_range; sub_mean -> head -> groups of RCABs -> +head -> upsampler ->
    tail -> add_mean (rcan_model.py:149-159)."""

    def __init__(self, scale, n_resgroups=10, n_resblocks=20, n_feats=64, reduction=16, res_scale=1.0,
                 rgb_range=255.0, *, compute_dtype=None, trunk_f32=True, device=None):
        super().__init__(scale, n_resgroups, n_resblocks, n_feats, reduction, res_scale, True, rgb_range,
                         dict(compute_dtype=compute_dtype, trunk_f32=trunk_f32, device=device))


class RCANCore(_RCANBase):
    """RCAN without MeanShift; input already mean-subtracted (rcan_model.py:166-198)."""

    def __init__(self, scale, n_resgroups=10, n_resblocks=20, n_feats=64, reduction=16, res_scale=1.0,
                 *, compute_dtype=None, trunk_f32=True, device=None):
        super().__init__(scale, n_resgroups, n_resblocks, n_feats, reduction, res_scale, False, 255.0,
                         dict(compute_dtype=compute_dtype, trunk_f32=trunk_f32, device=device))


def load_rcan_from_checkpoint(checkpoint_path, scale, **engine_kwargs):
    """RCAN(scale) with default sizes; accepts a bare state_dict or one under 'state_dict' /
    'model', strips a 'module.' prefix, strict=True (rcan_model.py:205-227)."""
    ckpt = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
    if isinstance(ckpt, dict):
        for c in ("state_dict", "model"):
            if c in ckpt:
                ckpt = ckpt[c]
                break
    sd = {(k[7:] if k.startswith("module.") else k): v for k, v in ckpt.items()}
    model = RCAN(scale=scale, **engine_kwargs)
    model.load_state_dict(sd, strict=True)
    return model


def create_core_model_from_full(full_model, scale):
    """RCANCore carrying the full model's weights minus MeanShift (rcan_model.py:230-253)."""
    core = RCANCore(scale=scale, n_resgroups=full_model.n_resgroups, n_resblocks=full_model.n_resblocks,
                    n_feats=full_model.n_feats, reduction=full_model.reduction, res_scale=full_model.res_scale,
                    compute_dtype=full_model._srmi_dtype, trunk_f32=full_model._srmi_trunk_f32,
                    device=full_model._srmi_device)
    full = full_model.state_dict()
    sd = core.state_dict()
    matched = 0
    for k in sd:
        if k in full:
            sd[k].copy_(full[k])
            matched += 1
        else:
            print(f"  warning: core key '{k}' is missing from the full model")
    print(f"  copied {matched}/{len(sd)} tensors")
    return core


def get_meanshift_params():
    return {"rgb_mean": [0.4488, 0.4371, 0.4040], "rgb_std": [1.0, 1.0, 1.0], "rgb_range": 255.0}


def infer_scale_from_model_name(model_path):
    """Scale from the file name: bix2/x2 -> 2, bix3/bdx3/x3 -> 3, bix4/x4 -> 4, bix8/x8 -> 8,
    default 4 (rcan_model.py:268-279)."""
    name = model_path.lower()
    for needle, scale in (("x2", 2), ("x3", 3), ("x4", 4), ("x8", 8)):
        if needle in name:
            return scale
    return 4
