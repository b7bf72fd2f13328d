#!/usr/bin/env python3
"""Generate tests/golden/prepost_cpp_ref.npz with the reference C++ runners' OWN pre-processing, post-processing
and bilinear resize functions (build container only).  `make -C oracle ref` compiles the reference's edsr.cpp /
rcan.cpp / realesrgan.cpp (+ its neuron_executor.cpp) from where they lie under /root/reference into oracle/_ref/;
oracle/ref_prepost_driver.cpp includes the runner source so that its file-local functions can be called:
  edsr.cpp:21-41, rcan.cpp:21-41, realesrgan.cpp:28-48       preprocess_image
  edsr.cpp:46-94, rcan.cpp:46-94, realesrgan.cpp:53-101      resize_image
  edsr.cpp:101-122, rcan.cpp:101-122, realesrgan.cpp:110-130 postprocess_image  ((uint8)(v + 0.5f): half-up)
Inputs: a seeded u8 image, the tie tensor of tools/make_golden_prepost.py (exact .5 ties, floats one ulp either side,
negatives, values beyond the range) and seeded images for six resize shapes.  Re-run: python tools/make_golden_prepost_cpp.py
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from superresolution_amd import synth  # noqa: E402
from make_golden_prepost import tie_tensor  # noqa: E402

RESIZE_CASES = [(60, 90, 113, 170), (113, 170, 60, 90), (5, 7, 3, 2), (16, 16, 16, 16), (1, 1, 4, 4), (200, 300, 339, 510)]
RGB_MEAN = np.asarray((0.4488, 0.4371, 0.4040), np.float32)


def load(name):
    L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", f"libref_{name}.so"))
    L.ref_preprocess.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_float]
    L.ref_postprocess.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_float]
    L.ref_resize.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
    return L


def p(a):
    return a.ctypes.data_as(C.c_void_p)


def run_all(L, img, t, mean, rng):
    """(preprocessed image, post-processed tie tensor) of one runner."""
    h, w, _ = img.shape
    pre = np.empty((3, h, w), np.float32)
    L.ref_preprocess(p(img), p(pre), h, w, p(mean), rng)
    _, th, tw = t.shape
    post = np.empty((th, tw, 3), np.uint8)
    L.ref_postprocess(p(np.ascontiguousarray(t)), p(post), th, tw, p(mean), rng)
    return pre, post


def main():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True)
    libs = {n: load(n) for n in ("edsr", "rcan", "realesrgan")}
    img = synth.random_image(21, 37, 77)
    t = tie_tensor()[0]                      # 0..255 domain
    zero = np.zeros(3, np.float32)
    out = {"img": img, "t": t, "rgb_mean": RGB_MEAN}
    for n in ("edsr", "rcan"):
        out[f"pre_{n}"], out[f"post_{n}_mean"] = run_all(libs[n], img, t, RGB_MEAN, 255.0)
        _, out[f"post_{n}_zero"] = run_all(libs[n], img, t, zero, 255.0)   # ties stay ties
    t_unit = (t * np.float32(1.0 / 255.0)).astype(np.float32)
    out["t_unit"] = t_unit
    out["pre_realesrgan"], out["post_realesrgan"] = run_all(libs["realesrgan"], img, t_unit, zero, 0.0)
    for i, (ih, iw, oh, ow) in enumerate(RESIZE_CASES):
        src = synth.random_image(ih, iw, 300 + i)
        res = {}
        for n, L in libs.items():
            dst = np.empty((oh, ow, 3), np.uint8)
            assert L.ref_resize(p(src), iw, ih, p(dst), ow, oh) == 0
            res[n] = dst
        assert np.array_equal(res["edsr"], res["rcan"]) and np.array_equal(res["edsr"], res["realesrgan"])
        out[f"resize_{i}"] = res["edsr"]
    out["resize_cases"] = np.asarray(RESIZE_CASES, np.int32)
    path = os.path.join(ROOT, "tests", "golden", "prepost_cpp_ref.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
