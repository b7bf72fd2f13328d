"""Runs code compiled from the reference's own C++ sources (oracle/_ref, `make -C oracle ref`, build container only) in a
short-lived process of its own, so that the test runner never maps reference-derived libraries (their static
initialisers run in whatever process loads them).  Not collected by pytest; started by tests/test_oracle.py and
tests/test_runner_errors_vs_reference.py.

  python tests/ref_cpp_check.py prepost              -> exit 0 when the built functions equal the numpy oracle
  python tests/ref_cpp_check.py errors <m> <tmpdir>  -> JSON {"ref": {...}, "mine": {...}} of error-path return codes
"""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import sr_oracle as O  # noqa: E402
from superresolution_amd import synth  # noqa: E402


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def prepost():
    """oracle/_ref against the numpy oracle on inputs the committed fixture does not hold: random images, random float
    tensors, 40 random resize shapes per runner."""
    rng = np.random.default_rng(5)
    mean = np.asarray(O.RGB_MEAN, np.float32)
    for name in ("edsr", "rcan", "realesrgan"):
        L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", f"libref_{name}.so"))
        L.ref_preprocess.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_float]
        L.ref_postprocess.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_float]
        L.ref_resize.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
        unit = name == "realesrgan"
        for k in range(6):
            h, w = int(rng.integers(1, 70)), int(rng.integers(1, 90))
            img = synth.random_image(h, w, 400 + k)
            pre = np.empty((3, h, w), np.float32)
            L.ref_preprocess(_p(img), _p(pre), h, w, _p(mean), 255.0)
            assert np.array_equal(pre, (O.pre_unit(img) if unit else O.pre_meanshift(img))[0].numpy())
            y = (synth.hash_uniform("fresh", (3, h, w), 500 + k) * (0.7 if unit else 180.0) + (0.5 if unit else 100.0)).astype(np.float32)
            u = np.empty((h, w, 3), np.uint8)
            L.ref_postprocess(_p(y), _p(u), h, w, _p(mean), 255.0)
            yt = torch.from_numpy(y)[None]
            want = O.post_unit(yt, O.ROUND_HALF_UP) if unit else O.post_range(yt, O.ROUND_HALF_UP, add=mean * np.float32(255.0))
            assert np.array_equal(u, want[0])
        for k in range(40):
            ih, iw, oh, ow = (int(v) for v in rng.integers(1, 160, 4))
            src = synth.random_image(ih, iw, 600 + k)
            dst = np.empty((oh, ow, 3), np.uint8)
            assert L.ref_resize(_p(src), iw, ih, _p(dst), ow, oh) == 0
            assert np.array_equal(dst, O.resize_bilinear_u8(src, ow, oh)), (name, ih, iw, oh, ow)
    print("reference C++ pre/post/resize equal the oracle")


def _bind(L, m):
    f = {}
    for name, args in (("init", [C.c_char_p, C.c_void_p]), ("release", [C.c_void_p]),
                       ("inference", [C.c_void_p, C.c_void_p, C.c_void_p]),
                       ("inference_ex", [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p])):
        fn = getattr(L, f"{m}_{name}")
        fn.argtypes, fn.restype = args, C.c_int
        f[name] = fn
    return f


def _calls(f, tmp_dir):
    """Return codes of the safe error-path calls, in a fixed order."""
    ctx = C.create_string_buffer(512)           # zero-filled: an uninitialised context
    img = C.create_string_buffer(4 * 4 * 3)
    out = C.c_void_p(0)
    ow, oh = C.c_int(0), C.c_int(0)
    garbage = os.path.join(tmp_dir, "garbage.bin")
    with open(garbage, "wb") as g:
        g.write(b"not a model" * 10)
    rc = {}
    rc["init(NULL, ctx)"] = f["init"](None, ctx)
    rc["init(path, NULL)"] = f["init"](b"/nonexistent/model.bin", None)
    rc["init(missing file, ctx)"] = f["init"](b"/nonexistent/model.bin", ctx)
    rc["ctx still zero after failed init"] = int(ctx.raw == bytes(512))
    rc["init(garbage file, ctx)"] = f["init"](garbage.encode(), ctx)
    rc["release(NULL)"] = f["release"](None)
    ctx = C.create_string_buffer(512)
    rc["release(uninitialised ctx)"] = f["release"](ctx)
    rc["inference(uninitialised ctx)"] = f["inference"](ctx, img, C.byref(out))
    rc["inference_ex(NULL ctx)"] = f["inference_ex"](None, img, 4, 4, C.byref(out), C.byref(ow), C.byref(oh))
    rc["inference_ex(uninitialised ctx)"] = f["inference_ex"](ctx, img, 4, 4, C.byref(out), C.byref(ow), C.byref(oh))
    rc["inference_ex(uninitialised ctx, NULL sizes)"] = f["inference_ex"](ctx, img, 4, 4, C.byref(out), None, None)
    rc["output pointer untouched"] = int(not out.value)
    return rc


def errors(m, tmp_dir):
    from superresolution_amd import _lib
    ref = _bind(C.CDLL(os.path.join(ROOT, "oracle", "_ref", f"libref_{m}.so")), m)
    mine = _bind(_lib.lib(), m)
    # the libraries print their diagnostics on stderr / stdout: the JSON goes out last, on one line of its own
    res = {"ref": _calls(ref, tmp_dir), "mine": _calls(mine, tmp_dir)}
    sys.stdout.flush()
    print("\nRESULT " + json.dumps(res))


if __name__ == "__main__":
    if sys.argv[1] == "prepost":
        prepost()
    else:
        errors(sys.argv[2], sys.argv[3])
