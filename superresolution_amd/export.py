"""Weight files for the C runner API (include/edsr.h, rcan.h, realesrgan.h).

The reference runners load a vendor-compiled `.dla` blob (edsr.cpp:126-157); the MI355X runners
load this container instead: the architecture hyper-parameters plus the state_dict tensors in the
reference's key layout, float32 little endian ("SRMIW001", see csrc/runners.cpp).  A model calibrated for the int8
mode, and with that mode switched on when it is saved, carries its activation ranges in an optional trailer
("SRMIQ001", int32 count, float32 values): the runner then starts in w8a8 mode, as the reference's runner loads
an already quantised .rknn file.  A model running in f16 when it is saved comes back in f16.
"""
from __future__ import annotations

import struct

import numpy as np
import torch

from . import _lib


def _desc(model):
    cfg = model._engine_config()
    return (cfg["arch"], cfg["scale"], cfg["n_feats"], cfg["n_blocks"], cfg.get("n_groups", 0),
            cfg.get("grow_ch", 0), cfg.get("reduction", 0), int(cfg.get("with_meanshift", False)),
            float(cfg.get("res_scale", 1.0)))


def save_weights(model, path):
    """Write `model` (an EDSR / EDSRCore / RCAN / RCANCore / RRDBNet from this package) to `path`."""
    arch, scale, nf, nb, ng, gc, red, ms, rs = _desc(model)
    sd = model.state_dict()
    with open(path, "wb") as f:
        f.write(b"SRMIW001")
        f.write(struct.pack("<8i", arch, scale, nf, nb, ng, gc, red, ms))
        f.write(struct.pack("<f", rs))
        f.write(struct.pack("<i", len(sd)))
        for k, v in sd.items():
            a = np.ascontiguousarray(v.detach().to("cpu", torch.float32).numpy())
            name = k.encode()
            f.write(struct.pack("<i", len(name)))
            f.write(name)
            f.write(struct.pack("<i", a.ndim))
            f.write(struct.pack(f"<{a.ndim}q", *a.shape))
            f.write(a.tobytes())
        # activation ranges of a calibrated model whose int8 form is switched ON (only looked up when an engine
        # already exists): a reader of the trailer starts in w8a8 mode, so a model that was calibrated and then
        # switched back to f16 (set_w8a8(False)) is written without it and comes back in f16
        eng = getattr(model, "_srmi_engine", None)
        live = eng is not None and getattr(model, "_srmi_stamp", None) == model._param_stamp()
        cal = eng.calibration() if live and eng.w8a8() else ()
        if len(cal):
            f.write(b"SRMIQ001")
            f.write(struct.pack("<i", len(cal)))
            f.write(np.asarray(cal, dtype="<f4").tobytes())
    return path


def convert_checkpoint(checkpoint_path, out_path, arch=None, scale=None):
    """Official checkpoint (.pt/.pth) -> runner weight file.  `arch` defaults to a guess from the
    file name (edsr / rcan / anything else = Real-ESRGAN)."""
    from .edsr_model import load_edsr_from_checkpoint
    from .rcan_model import infer_scale_from_model_name, load_rcan_from_checkpoint
    from .realesrgan_model import load_realesrgan_from_checkpoint
    name = checkpoint_path.lower()
    arch = arch or ("edsr" if "edsr" in name else "rcan" if "rcan" in name else "realesrgan")
    if arch == "edsr":
        model = load_edsr_from_checkpoint(checkpoint_path, scale or 4)
    elif arch == "rcan":
        model = load_rcan_from_checkpoint(checkpoint_path, scale or infer_scale_from_model_name(checkpoint_path))
    else:
        model, _ = load_realesrgan_from_checkpoint(checkpoint_path)
    return save_weights(model, out_path)


def load_weights(path, **engine_kwargs):
    """Read a runner weight file back into the matching module of this package (EDSR / EDSRCore / RCAN /
    RCANCore / RRDBNet), weights loaded.  engine_kwargs: compute_dtype, trunk_f32, device, ..."""
    from .edsr_model import EDSR, EDSRCore
    from .rcan_model import RCAN, RCANCore
    from .realesrgan_model import RRDBNet
    with open(path, "rb") as f:
        if f.read(8) != b"SRMIW001":
            raise ValueError(f"{path} is not an SRMIW001 weight file")
        arch, scale, nf, nb, ng, gc, red, ms = struct.unpack("<8i", f.read(32))
        (rs,) = struct.unpack("<f", f.read(4))
        (nt,) = struct.unpack("<i", f.read(4))
        sd = {}
        for _ in range(nt):
            (nl,) = struct.unpack("<i", f.read(4))
            name = f.read(nl).decode()
            (nd,) = struct.unpack("<i", f.read(4))
            dims = struct.unpack(f"<{nd}q", f.read(8 * nd))
            n = int(np.prod(dims)) if nd else 1
            sd[name] = torch.from_numpy(np.frombuffer(f.read(4 * n), dtype="<f4").reshape(dims).copy())
        cal = None
        if f.read(8) == b"SRMIQ001":
            (nc,) = struct.unpack("<i", f.read(4))
            cal = np.frombuffer(f.read(4 * nc), dtype="<f4").copy()
    if arch == _lib.ARCH_RRDBNET:
        model = RRDBNet(num_feat=nf, num_block=nb, num_grow_ch=gc, scale=scale, **engine_kwargs)
    elif arch == _lib.ARCH_EDSR:
        cls = EDSR if ms else EDSRCore
        model = cls(n_resblocks=nb, n_feats=nf, scale=scale, res_scale=rs, **engine_kwargs)
    elif arch == _lib.ARCH_RCAN:
        cls = RCAN if ms else RCANCore
        model = cls(scale=scale, n_resgroups=ng, n_resblocks=nb, n_feats=nf, reduction=red, res_scale=rs, **engine_kwargs)
    else:
        raise ValueError(f"{path}: unknown architecture {arch}")
    model.load_state_dict(sd, strict=True)
    if cal is not None:
        model.engine().set_calibration(cal)
    return model
