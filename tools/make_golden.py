#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the *reference's own modules* (imported from
/root/reference, which exists only in the build container) on deterministic synthetic
state_dicts and inputs from superresolution_amd.synth.

Each fixture stores the generating parameters (arch kwargs, seed, input spec) and the reference
output; tests regenerate weights and inputs from the same parameters, so no weights are stored.
Re-run:  python tools/make_golden.py
"""
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from superresolution_amd import synth  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _sd_torch(sd):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}


def input_tensor(kind, n, h, w, seed):
    """kind 'unit': u8/255 in [0,1]; 'range': u8 as float 0..255."""
    img = synth.random_image(h, w, seed, n=n).astype(np.float32)
    if kind == "unit":
        img = img * np.float32(1.0 / 255.0)
    return torch.from_numpy(img).permute(0, 3, 1, 2).contiguous()


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    rrdb = _load(f"{REF}/realesrgan/mtk/python/realesrgan_model.py", "ref_realesrgan_model")
    edsr = _load(f"{REF}/edsr/mtk/python/edsr_model.py", "ref_edsr_model")
    rcan = _load(f"{REF}/rcan/mtk/python/rcan_model.py", "ref_rcan_model")

    def run(model, x):
        model.eval()
        with torch.no_grad():
            return model(x).numpy()

    cases = []
    # ---- RRDBNet ------------------------------------------------------------------------------
    for tag, kw, (n, h, w), seed in [
        ("rrdbnet_b23_x4_16x16", dict(num_block=23, scale=4), (1, 16, 16), 1),
        ("rrdbnet_b1_x4_33x47", dict(num_block=1, scale=4), (1, 33, 47), 2),
        ("rrdbnet_b2_x2_20x24_n2", dict(num_block=2, scale=2), (2, 20, 24), 3),
        ("rrdbnet_b6_x4_24x40", dict(num_block=6, scale=4), (1, 24, 40), 4),
    ]:
        sd = synth.rrdbnet_state_dict(seed=seed, **kw)
        m = rrdb.RRDBNet(num_in_ch=3, num_out_ch=3, num_feat=64, num_grow_ch=32, **kw)
        m.load_state_dict(_sd_torch(sd), strict=True)
        x = input_tensor("unit", n, h, w, seed)
        y = run(m, x)
        cases.append((tag, dict(arch="rrdbnet", seed=seed, n=n, h=h, w=w, input="unit", **kw), y))

    # ---- EDSR ---------------------------------------------------------------------------------
    for tag, kw, (n, h, w), seed in [
        ("edsr_r32f256_x4_16x16", dict(n_resblocks=32, n_feats=256, scale=4), (1, 16, 16), 5),
        ("edsr_r2f64_x4_33x47", dict(n_resblocks=2, n_feats=64, scale=4), (1, 33, 47), 6),
        ("edsr_r2f64_x3_17x19", dict(n_resblocks=2, n_feats=64, scale=3), (1, 17, 19), 7),
        ("edsr_r3f64_x2_20x24_n2", dict(n_resblocks=3, n_feats=64, scale=2), (2, 20, 24), 8),
    ]:
        sd = synth.edsr_state_dict(seed=seed, **kw)
        m = edsr.EDSR(**kw)
        m.load_state_dict(_sd_torch(sd), strict=True)
        x = input_tensor("range", n, h, w, seed)
        y = run(m, x)
        # the reference's own identity: full(x) == core(x - 255*mean) + 255*mean
        # (edsr/mtk/python/test/test_pytorch.py:137-198, tolerance 1e-3)
        core = edsr.EDSRCore(**kw)
        core.load_state_dict({k: v for k, v in m.state_dict().items() if "mean" not in k})
        mean = torch.tensor(synth.RGB_MEAN).view(1, 3, 1, 1) * 255.0
        yc = run(core, x - mean) + mean.numpy()
        assert np.abs(yc - y).max() < 1e-3, np.abs(yc - y).max()
        cases.append((tag, dict(arch="edsr", seed=seed, n=n, h=h, w=w, input="range",
                                res_scale=0.1, **kw), y))

    # ---- RCAN ---------------------------------------------------------------------------------
    for tag, kw, (n, h, w), seed in [
        ("rcan_g10b20_x4_16x16", dict(scale=4, n_resgroups=10, n_resblocks=20), (1, 16, 16), 9),
        ("rcan_g1b2_x4_33x47", dict(scale=4, n_resgroups=1, n_resblocks=2), (1, 33, 47), 10),
        ("rcan_g2b2_x3_17x19", dict(scale=3, n_resgroups=2, n_resblocks=2), (1, 17, 19), 11),
        ("rcan_g2b3_x2_20x24_n2", dict(scale=2, n_resgroups=2, n_resblocks=3), (2, 20, 24), 12),
        ("rcan_g1b1_x8_9x11", dict(scale=8, n_resgroups=1, n_resblocks=1), (1, 9, 11), 13),
    ]:
        sd = synth.rcan_state_dict(seed=seed, **kw)
        m = rcan.RCAN(**kw)
        m.load_state_dict(_sd_torch(sd), strict=True)
        x = input_tensor("range", n, h, w, seed)
        y = run(m, x)
        cases.append((tag, dict(arch="rcan", seed=seed, n=n, h=h, w=w, input="range",
                                n_feats=64, reduction=16, res_scale=1.0, **kw), y))

    for tag, meta, y in cases:
        np.savez_compressed(os.path.join(OUT, tag + ".npz"), y=y.astype(np.float32),
                            meta=np.array(repr(meta)))
        print(f"{tag:32s} out {tuple(y.shape)} range [{y.min():.3f}, {y.max():.3f}] "
              f"mean {y.mean():.3f} std {y.std():.3f}")

    # ---- per-op fixtures (reference building blocks) --------------------------------------------
    ops = {}
    # ResidualDenseBlock / RRDB (realesrgan_model.py:13-49)
    sd = synth.rrdbnet_state_dict(num_block=1, seed=21)
    blk = rrdb.RRDB(64, 32)
    blk.load_state_dict(_sd_torch({k[len("body.0."):]: v for k, v in sd.items()
                                   if k.startswith("body.0.")}))
    x = torch.from_numpy(synth.hash_uniform("rrdb_in", (1, 64, 13, 21), 21))
    ops["rrdb_in_key"] = np.array("rrdb_in")
    ops["rrdb_out"] = run(blk, x)
    ops["rdb1_out"] = run(blk.rdb1, x)
    # CALayer / RCAB (rcan_model.py:61-97)
    sd = synth.rcan_state_dict(n_resgroups=1, n_resblocks=1, seed=22)
    rc = rcan.RCAB(rcan.default_conv, 64, 3, 16)
    rc.load_state_dict(_sd_torch({k[len("body.0.body.0."):]: v for k, v in sd.items()
                                  if k.startswith("body.0.body.0.")}))
    x = torch.from_numpy(synth.hash_uniform("rcab_in", (2, 64, 11, 15), 22)) * 50.0
    ops["rcab_out"] = run(rc, x)
    ops["ca_out"] = run(rc.body[3], x)
    # PixelShuffle, nearest (torch ops the reference calls)
    x = torch.from_numpy(synth.hash_uniform("ps_in", (1, 36, 5, 7), 23))
    ops["ps2_out"] = torch.nn.PixelShuffle(2)(x).numpy()
    ops["ps3_out"] = torch.nn.PixelShuffle(3)(x).numpy()
    ops["near_out"] = torch.nn.functional.interpolate(x, scale_factor=2, mode="nearest").numpy()
    # ResBlock (edsr_model.py:29-47)
    sd = synth.edsr_state_dict(n_resblocks=1, n_feats=64, seed=24)
    rb = edsr.ResBlock(edsr.default_conv, 64, 3, res_scale=0.1)
    rb.load_state_dict(_sd_torch({k[len("body.0."):]: v for k, v in sd.items()
                                  if k.startswith("body.0.")}))
    x = torch.from_numpy(synth.hash_uniform("rb_in", (1, 64, 9, 10), 24)) * 100.0
    ops["resblock_out"] = run(rb, x)
    # filename heuristics (realesrgan_model.py:126-139, rcan_model.py:268-279)
    names = ["RealESRGAN_x4plus.pth", "RealESRGAN_x4plus_anime_6B.pth", "RealESRNet_x4plus.pth",
             "RealESRGAN_x2plus.pth", "foo.pth", "RCAN_BIX4.pt", "RCAN_BIX2.pt", "RCAN_BIX3.pt",
             "RCAN_BIX8.pt", "RCAN_BDX3.pt", "weights.pt"]
    ops["names"] = np.array(names)
    ops["infer_model_type"] = np.array([repr(rrdb.infer_model_type(n)) for n in names])
    ops["infer_scale"] = np.array([rcan.infer_scale_from_model_name(n) for n in names])
    ops["edsr_meanshift_params"] = np.array(repr(edsr.get_meanshift_params()))
    ops["rcan_meanshift_params"] = np.array(repr(rcan.get_meanshift_params()))
    # parameter counts / key lists of the default nets
    for nm, mod in [("rrdbnet", rrdb.RRDBNet()), ("edsr", edsr.EDSR()), ("rcan", rcan.RCAN(scale=4))]:
        sdm = mod.state_dict()
        ops[f"{nm}_keys"] = np.array(list(sdm.keys()))
        ops[f"{nm}_shapes"] = np.array([repr(tuple(v.shape)) for v in sdm.values()])
    np.savez_compressed(os.path.join(OUT, "ops.npz"), **ops)
    print("ops.npz written:", sorted(ops.keys()))


if __name__ == "__main__":
    main()
