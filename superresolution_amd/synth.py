"""Deterministic synthetic weights and images.

The official checkpoints (EDSR_x4.pt, RCAN_BIX4.pt, RealESRGAN_x4plus.pth) are not
redistributable with this repository, so tests and benchmarks use state_dicts generated here in
the exact key/shape layout of the reference modules
(/root/reference/edsr/mtk/python/edsr_model.py:80-124,
 /root/reference/rcan/mtk/python/rcan_model.py:119-159,
 /root/reference/realesrgan/mtk/python/realesrgan_model.py:60-123).

Values come from a counter-based integer hash (splitmix64 finaliser) of (seed, crc32(key), index),
so they are bit-identical on every machine and independent of torch's RNG implementation.
Only numpy is required.
"""
from __future__ import annotations

import zlib
from collections import OrderedDict

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint64, copy=True)
    with np.errstate(over="ignore"):
        x += np.uint64(0x9E3779B97F4A7C15)
        x ^= x >> np.uint64(30)
        x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27)
        x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    return x


def hash_uniform(key: str, shape, seed: int = 0) -> np.ndarray:
    """float32 array of `shape`, uniform in [-1, 1), a pure function of (key, seed, index)."""
    n = int(np.prod(shape)) if len(shape) else 1
    base = (np.uint64(zlib.crc32(key.encode())) << np.uint64(32)) ^ np.uint64(seed & 0xFFFFFFFF)
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) + _splitmix64(np.array([base], dtype=np.uint64))[0]
    bits = _splitmix64(idx) >> np.uint64(40)  # 24 random bits
    u = bits.astype(np.float64) * (1.0 / (1 << 24))  # [0,1)
    return (u * 2.0 - 1.0).astype(np.float32).reshape(shape)


def _conv(sd, name, cout, cin, k, gain, seed, bias_amp=0.02):
    fan_in = cin * k * k
    a = gain * np.sqrt(3.0 / fan_in)
    sd[name + ".weight"] = hash_uniform(name + ".weight", (cout, cin, k, k), seed) * np.float32(a)
    sd[name + ".bias"] = hash_uniform(name + ".bias", (cout,), seed) * np.float32(bias_amp)


RGB_MEAN = (0.4488, 0.4371, 0.4040)


def _meanshift(sd, rgb_range):
    # Values the official EDSR/RCAN checkpoints carry for the frozen MeanShift layers
    # (identity 1x1 weight, bias = -/+ rgb_range * mean; edsr_model.py:18-26).
    eye = np.eye(3, dtype=np.float32).reshape(3, 3, 1, 1)
    m = np.asarray(RGB_MEAN, dtype=np.float32) * np.float32(rgb_range)
    sd["sub_mean.weight"] = eye.copy()
    sd["sub_mean.bias"] = -m
    sd["add_mean.weight"] = eye.copy()
    sd["add_mean.bias"] = m.copy()


def _upsampler(sd, prefix, scale, nf, seed, gain):
    if scale & (scale - 1) == 0:
        n = int(round(np.log2(scale)))
        for i in range(n):
            _conv(sd, f"{prefix}.{2 * i}", 4 * nf, nf, 3, gain, seed)
    elif scale == 3:
        _conv(sd, f"{prefix}.0", 9 * nf, nf, 3, gain, seed)
    else:
        raise NotImplementedError(f"scale {scale}")


def rrdbnet_state_dict(num_in_ch=3, num_out_ch=3, num_feat=64, num_block=23, num_grow_ch=32,
                       scale=4, seed=0):
    """State dict in the key layout of RRDBNet (realesrgan_model.py:67-94)."""
    sd = OrderedDict()
    nf, gc = num_feat, num_grow_ch
    _conv(sd, "conv_first", nf, num_in_ch, 3, 1.6, seed, 0.05)
    for i in range(num_block):
        for r in (1, 2, 3):
            p = f"body.{i}.rdb{r}"
            for j in range(4):
                _conv(sd, f"{p}.conv{j + 1}", gc, nf + j * gc, 3, 1.0, seed)
            _conv(sd, f"{p}.conv5", nf, nf + 4 * gc, 3, 0.45, seed)
    # every RRDB returns 0.2*out + x with out ~ x, i.e. the trunk grows ~1.2x per block whatever the
    # branch weights are; conv_body undoes that so the upsampling stages see O(1) activations.
    _conv(sd, "conv_body", nf, nf, 3, 0.7 / 1.2 ** num_block, seed)
    _conv(sd, "conv_up1", nf, nf, 3, 1.3, seed)
    if scale == 4:
        _conv(sd, "conv_up2", nf, nf, 3, 1.3, seed)
    _conv(sd, "conv_hr", nf, nf, 3, 1.3, seed)
    _conv(sd, "conv_last", num_out_ch, nf, 3, 0.45, seed, 0.0)
    # Centre the output in the displayable range so that clamping and rounding are both exercised.
    sd["conv_last.bias"] = np.full((num_out_ch,), 0.5, dtype=np.float32) + \
        hash_uniform("conv_last.bias", (num_out_ch,), seed) * np.float32(0.05)
    return sd


def edsr_state_dict(n_resblocks=32, n_feats=256, scale=4, rgb_range=255, seed=0):
    """State dict in the key layout of EDSR (edsr_model.py:82-112); res_scale is not a parameter."""
    sd = OrderedDict()
    _meanshift(sd, rgb_range)
    _conv(sd, "head.0", n_feats, 3, 3, 1.0, seed, 0.5)
    for i in range(n_resblocks):
        _conv(sd, f"body.{i}.body.0", n_feats, n_feats, 3, 1.4, seed, 0.5)
        _conv(sd, f"body.{i}.body.2", n_feats, n_feats, 3, 1.4, seed, 0.5)
    _conv(sd, f"body.{n_resblocks}", n_feats, n_feats, 3, 0.7, seed, 0.5)
    _upsampler(sd, "tail.0", scale, n_feats, seed, 1.0)
    _conv(sd, "tail.1", 3, n_feats, 3, 0.5, seed, 0.5)
    return sd


def rcan_state_dict(scale=4, n_resgroups=10, n_resblocks=20, n_feats=64, reduction=16,
                    rgb_range=255.0, seed=0):
    """State dict in the key layout of RCAN (rcan_model.py:123-147), MeanShift weights = identity
    as in the official checkpoint (the module's own initialiser leaves them random,
    rcan_model.py:28-34)."""
    sd = OrderedDict()
    _meanshift(sd, rgb_range)
    nf = n_feats
    _conv(sd, "head.0", nf, 3, 3, 1.0, seed, 0.5)
    for g in range(n_resgroups):
        for b in range(n_resblocks):
            p = f"body.{g}.body.{b}.body"
            _conv(sd, f"{p}.0", nf, nf, 3, 1.4, seed, 0.5)
            _conv(sd, f"{p}.2", nf, nf, 3, 0.25, seed, 0.5)
            # channel attention squeeze/excite; inputs are pooled means in the 0..255 domain, so
            # the squeeze weights are small to keep the sigmoid away from saturation.
            _conv(sd, f"{p}.3.conv_du.0", nf // reduction, nf, 1, 0.05, seed, 0.1)
            _conv(sd, f"{p}.3.conv_du.2", nf, nf // reduction, 1, 0.5, seed, 0.5)
        _conv(sd, f"body.{g}.body.{n_resblocks}", nf, nf, 3, 0.3, seed, 0.5)
    _conv(sd, f"body.{n_resgroups}", nf, nf, 3, 0.7, seed, 0.5)
    _upsampler(sd, "tail.0", scale, nf, seed, 1.0)
    _conv(sd, "tail.1", 3, nf, 3, 0.5, seed, 0.5)
    return sd


def random_image(h, w, seed=0, n=None):
    """Seeded uniform uint8 RGB image(s), HWC (or NHWC when n is given)."""
    shape = (h, w, 3) if n is None else (n, h, w, 3)
    return np.random.default_rng(seed).integers(0, 256, shape, dtype=np.uint8)


def smooth_image(h, w, seed=0):
    """Smooth gradient + low-amplitude noise RGB image, so that the network output is image-like
    (same idea as the reference's synthetic calibration pattern,
    /root/reference/edsr/rknn/python/prepare_quantization_dataset.py:102-115)."""
    yy, xx = np.meshgrid(np.linspace(0, 1, h, dtype=np.float32),
                         np.linspace(0, 1, w, dtype=np.float32), indexing="ij")
    base = np.stack([
        0.5 + 0.4 * np.sin(6.0 * xx + 2.0 * yy),
        0.5 + 0.4 * np.cos(5.0 * yy - 1.0 * xx),
        0.2 + 0.6 * xx * yy + 0.2 * np.sin(9.0 * (xx - yy)),
    ], axis=-1)
    noise = hash_uniform(f"smooth{h}x{w}", (h, w, 3), seed) * 0.05
    return np.clip((base + noise) * 255.0, 0, 255).astype(np.uint8)
