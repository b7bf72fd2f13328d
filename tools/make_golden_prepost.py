#!/usr/bin/env python3
"""Generate tests/golden/prepost_ref.npz with the reference's OWN Python pre/post-processing functions, imported
from /root/reference (build container only):
  edsr/rknn/python/test_pytorch.py:109-124   preprocess (PIL -> f32 CHW 0..255), postprocess (clamp, round half-even, byte)
  rcan/rknn/python/test_pytorch.py:170-185   the same two functions of the RCAN script
  edsr/mtk/python/test/test_pytorch.py:47-64 load_image_tensor, save_image (clip, astype(uint8) = truncation)
The float tensor holds exact .5 ties (even and odd), values one ulp either side of a tie, negatives and values
beyond 255, so every branch of the rounding is pinned.  Re-run: python tools/make_golden_prepost.py
"""
import importlib.util
import os
import sys
import tempfile

import numpy as np
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from superresolution_amd import synth  # noqa: E402

REF = "/root/reference"


def _load(path, name):
    sys.path.insert(0, os.path.dirname(path))
    sys.path.insert(0, os.path.dirname(os.path.dirname(path)))   # mtk/python for `import edsr_model`
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def tie_tensor(h=24, w=40):
    n = 3 * h * w
    v = np.zeros(n, np.float32)
    ties = np.arange(0, 255, dtype=np.float32) + np.float32(0.5)          # 0.5 .. 254.5
    v[:255] = ties
    v[255:510] = np.nextafter(ties, np.float32(0))                          # just below a tie
    v[510:765] = np.nextafter(ties, np.float32(1000))                       # just above
    v[765:775] = [-3.5, -0.5, -0.49, -0.0, 0.0, 255.0, 255.4, 255.5, 256.5, 1e6]
    rest = synth.hash_uniform("prepost.rest", (n - 775,), 5) * 300.0 - 20.0  # [-20, 280)
    v[775:] = rest.astype(np.float32)
    return v.reshape(1, 3, h, w)


def main():
    edsr_rknn = _load(f"{REF}/edsr/rknn/python/test_pytorch.py", "ref_edsr_rknn_test")
    rcan_rknn = _load(f"{REF}/rcan/rknn/python/test_pytorch.py", "ref_rcan_rknn_test")
    edsr_mtk = _load(f"{REF}/edsr/mtk/python/test/test_pytorch.py", "ref_edsr_mtk_test")
    t = tie_tensor()
    out = {"t": t}
    out["post_half_even_edsr_rknn"] = np.array(edsr_rknn.postprocess(torch.from_numpy(t.copy())))
    out["post_half_even_rcan_rknn"] = np.array(rcan_rknn.postprocess(torch.from_numpy(t.copy())))
    img = synth.random_image(21, 37, 77)
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "o.png")
        assert edsr_mtk.save_image(torch.from_numpy(t.copy()), p)
        out["post_trunc_edsr_mtk"] = np.array(Image.open(p))
        q = os.path.join(d, "i.png")
        Image.fromarray(img, "RGB").save(q)
        out["img"] = img
        out["pre_edsr_rknn"] = edsr_rknn.preprocess(q)[0].numpy()
        out["pre_rcan_rknn"] = rcan_rknn.preprocess(q)[0].numpy()
        out["pre_edsr_mtk"] = edsr_mtk.load_image_tensor(q)[0].numpy()
    path = os.path.join(ROOT, "tests", "golden", "prepost_ref.npz")
    np.savez_compressed(path, **out)
    print("written", path, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
