"""Real-ESRGAN RRDBNet on MI355X — same construction, state_dict layout and forward(x) contract
as the reference module (/root/reference/realesrgan/mtk/python/realesrgan_model.py), with the
forward pass executed by HIP kernels through libsrmi355.so.

Names mirror the reference so `from superresolution_amd.realesrgan_model import RRDBNet,
load_realesrgan_from_checkpoint` is a drop-in:
  RRDBNet(num_in_ch, num_out_ch, num_feat, num_block, num_grow_ch, scale)   realesrgan_model.py:67
  infer_model_type(path)                                                     realesrgan_model.py:126
  load_realesrgan_from_checkpoint(path, device) -> (model, scale)            realesrgan_model.py:142
  RealESRGAN_x4plus / RealESRNet_x4plus / RealESRGAN_x4plus_anime_6B / RealESRGAN_x2plus  :199-220
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from . import _lib
from .engine import HipModule, conv_params


class ResidualDenseBlock(nn.Module):
    """Parameter container of one RDB: conv1..conv5 (realesrgan_model.py:16-22).  The math
    (x1..x4 = lrelu(conv_k(cat)), out = conv5(cat)*0.2 + x) runs fused inside the engine."""

    def __init__(self, num_feat=64, num_grow_ch=32):
        super().__init__()
        for j in range(4):
            setattr(self, f"conv{j + 1}", conv_params(num_feat + j * num_grow_ch, num_grow_ch, 3))
        self.conv5 = conv_params(num_feat + 4 * num_grow_ch, num_feat, 3)


class RRDB(nn.Module):
    """Parameter container of one residual-in-residual dense block (realesrgan_model.py:38-42)."""

    def __init__(self, num_feat, num_grow_ch=32):
        super().__init__()
        self.rdb1 = ResidualDenseBlock(num_feat, num_grow_ch)
        self.rdb2 = ResidualDenseBlock(num_feat, num_grow_ch)
        self.rdb3 = ResidualDenseBlock(num_feat, num_grow_ch)


class RRDBNet(HipModule):
    """x: f32 NCHW in [0,1] -> f32 NCHW (scale x larger), no clamp (realesrgan_model.py:96-123)."""

    def __init__(self, num_in_ch=3, num_out_ch=3, num_feat=64, num_block=23, num_grow_ch=32, scale=4,
                 *, compute_dtype=None, trunk_f32=True, device=None, growth_round8=True):
        super().__init__(compute_dtype, trunk_f32, device)
        # f16 compute only: the dense blocks' grown channels carry 8 significant bits (srmi_model_set_growth_round8)
        self._srmi_growth_round8 = bool(growth_round8)
        if scale not in (2, 4):
            raise ValueError(f"Unsupported scale: {scale}. Supported scales: 2, 4")
        if num_in_ch != 3 or num_out_ch != 3:
            raise NotImplementedError("the HIP engine implements the RGB networks (num_in_ch = num_out_ch = 3)")
        self.scale = scale
        self.num_feat, self.num_block, self.num_grow_ch = num_feat, num_block, num_grow_ch
        self.conv_first = conv_params(num_in_ch, num_feat, 3)
        self.body = nn.Sequential(*[RRDB(num_feat, num_grow_ch) for _ in range(num_block)])
        self.conv_body = conv_params(num_feat, num_feat, 3)
        self.conv_up1 = conv_params(num_feat, num_feat, 3)
        if scale == 4:
            self.conv_up2 = conv_params(num_feat, num_feat, 3)
        self.conv_hr = conv_params(num_feat, num_feat, 3)
        self.conv_last = conv_params(num_feat, num_out_ch, 3)

    def _engine_config(self):
        return dict(arch=_lib.ARCH_RRDBNET, scale=self.scale, n_feats=self.num_feat, n_blocks=self.num_block,
                    grow_ch=self.num_grow_ch, growth_round8=self._srmi_growth_round8)

    def quantize_w8a8(self, calibration_frames, batch=4):
        """Calibrate on `calibration_frames` (uint8 NHWC, or float32 NCHW in [0,1] like the reference's calibration
        .npy files, realesrgan/rknn/python/convert.py:198-213) and run the dense blocks in int8 from now on (f16
        compute only).  `self.engine().set_w8a8(False)` returns to plain f16.  Changing a parameter afterwards
        discards the calibration."""
        return self.engine().calibrate(calibration_frames, batch=batch)


def infer_model_type(model_path):
    """(type, scale, num_block) from the checkpoint file name, with the reference's precedence:
    anime -> 6 blocks; esrnet; x4; x2; default esrnet x4 (realesrgan_model.py:126-139)."""
    name = model_path.lower()
    for needle, result in (("anime", ("anime", 4, 6)), ("esrnet", ("esrnet", 4, 23)),
                           ("x4", ("esrgan", 4, 23)), ("x2", ("esrgan", 2, 23))):
        if needle in name:
            return result
    return "esrnet", 4, 23


def _unwrap(ckpt, containers):
    if isinstance(ckpt, dict):
        for c in containers:
            if c in ckpt:
                ckpt = ckpt[c]
                break
    return {(k[len("module."):] if k.startswith("module.") else k): v for k, v in ckpt.items()}


def load_realesrgan_from_checkpoint(checkpoint_path, device="cpu", **engine_kwargs):
    """Build the network the file name implies and load the weights strictly.  Accepts a bare
    state_dict or one nested under params_ema > params > model > state_dict, with or without a
    DataParallel 'module.' prefix (realesrgan_model.py:142-195).  Returns (model, scale)."""
    model_type, scale, num_block = infer_model_type(checkpoint_path)
    print(f"model type: {model_type}, scale: x{scale}, blocks: {num_block}")
    model = RRDBNet(num_in_ch=3, num_out_ch=3, num_feat=64, num_block=num_block, num_grow_ch=32,
                    scale=4 if model_type == "anime" else scale, **engine_kwargs)
    ckpt = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
    model.load_state_dict(_unwrap(ckpt, ("params_ema", "params", "model", "state_dict")), strict=True)
    model.eval()
    print(f"loaded {os.path.basename(checkpoint_path)} "
          f"({sum(p.numel() for p in model.parameters()) / 1e6:.2f}M parameters)")
    return model.to(device), scale


def RealESRGAN_x4plus(**kw):
    return RRDBNet(3, 3, 64, 23, 32, 4, **kw)


def RealESRNet_x4plus(**kw):
    return RRDBNet(3, 3, 64, 23, 32, 4, **kw)


def RealESRGAN_x4plus_anime_6B(**kw):
    return RRDBNet(3, 3, 64, 6, 32, 4, **kw)


def RealESRGAN_x2plus(**kw):
    return RRDBNet(3, 3, 64, 23, 32, 2, **kw)
