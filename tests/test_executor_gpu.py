"""The library used from compiled C and C++ programs that include the public headers (tests/cpp/):
Mi355Executor (include/mi355_executor.h), the stand-in for the reference's NeuronExecutor seam
(realesrgan/mtk/cpp/jni/src/mtk_npu/neuron_executor.h:85-117), and the runner API from plain C with the
headers' own context structs, including two contexts driven from two threads at once."""
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import sr_oracle as O
from superresolution_amd import _lib, edsr_model, export, synth
from tests import helpers as H

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "bin")


def _program(name):
    """Path of a test program; built ahead of time by __graft_entry__.build() (plain gcc / g++), here only if missing."""
    path = os.path.join(BIN, name)
    if not os.path.exists(path):
        subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "cpp")], check=True)
    return path


def _run(args):
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "superresolution_amd") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run(args, env=env, capture_output=True, text=True, timeout=170)
    assert r.returncode == 0, f"{args[0]} failed ({r.returncode}):\n{r.stdout}\n{r.stderr}"
    return r.stdout


@pytest.mark.parametrize("arch", ["rrdbnet", "edsr_core"])
def test_mi355_executor_from_cpp(tmp_path, arch, monkeypatch):
    monkeypatch.setenv("SRMI_DTYPE", "f32")
    if arch == "rrdbnet":
        meta = dict(arch="rrdbnet", num_block=2, scale=4, seed=61, input="unit")
        model, sd = H.build_model(meta, "f32")
        n, h, w = 2, 40, 56
        x = torch.from_numpy(synth.random_image(h, w, 9, n=n).astype(np.float32) / 255.0).permute(0, 3, 1, 2).contiguous()
        with torch.no_grad():
            ref = H.oracle_forward(meta, sd, x).numpy()
        tol = 1e-4
    else:
        # the network a .dla holds: EDSR without MeanShift (edsr_model.py:146-180); the runner subtracts the mean itself
        meta = dict(arch="edsr", n_resblocks=2, n_feats=64, scale=2, seed=62, input="range")
        full, sd = H.build_model(meta, "f32")
        model = edsr_model.create_core_model_from_full(full, 2)
        n, h, w = 1, 33, 47
        x = torch.from_numpy(synth.random_image(h, w, 10, n=n).astype(np.float32)).permute(0, 3, 1, 2).contiguous()
        mean = torch.tensor(edsr_model.get_meanshift_params()["rgb_mean"]).view(1, 3, 1, 1) * 255.0
        with torch.no_grad():
            ref = (H.oracle_forward(meta, sd, x) - mean).numpy()   # core(x - mean) = full(x) - mean
        x = x - mean
        tol = 1e-3
    s = model.engine().scale
    wpath, ipath, opath = (str(tmp_path / f) for f in ("model.srmi", "in.f32", "out.f32"))
    export.save_weights(model, wpath)
    x.numpy().tofile(ipath)
    out = _run([_program("executor_demo"), wpath, str(n), str(h), str(w), str(s), ipath, opath])
    assert out.startswith("ok")
    got = np.fromfile(opath, dtype=np.float32).reshape(n, 3, h * s, w * s)
    # same bytes as the Python module's forward on the same device path ...
    want = model(x).numpy()
    assert np.array_equal(got, want)
    # ... and the reference's numbers
    err = np.abs(got - ref).max()
    print(f"executor {arch}: max abs err vs oracle {err:.3g}")
    assert err <= tol


def test_runner_api_from_c_and_two_contexts_on_two_threads(tmp_path):
    metas = [dict(arch="rrdbnet", num_block=1, scale=4, seed=71, input="unit"),
             dict(arch="rrdbnet", num_block=2, scale=4, seed=72, input="unit")]
    img = synth.smooth_image(339, 510, 8)
    ipath = str(tmp_path / "in.u8")
    img.tofile(ipath)
    paths, wants = [], []
    for i, meta in enumerate(metas):
        model, _ = H.build_model(meta, "f16")
        p = str(tmp_path / f"RealESRGAN_x4plus_{i}.srmi")
        export.save_weights(model, p)
        paths.append(p)
        wants.append(model.forward_u8(img, _lib.ROUND_HALF_UP).numpy())
    # single context from C, the header's own struct
    o0 = str(tmp_path / "o0.u8")
    assert _run([_program("runner_demo"), "realesrgan", paths[0], ipath, "510", "339", o0]).startswith("ok 2040x1356")
    assert np.array_equal(np.fromfile(o0, dtype=np.uint8).reshape(1356, 2040, 3), wants[0])
    # two contexts, two threads, three inferences each at the same time: every result equals the single-context one
    oa, ob = str(tmp_path / "oa.u8"), str(tmp_path / "ob.u8")
    assert _run([_program("runner_demo"), "threads", paths[0], paths[1], ipath, "510", "339", oa, ob]).startswith("ok threads")
    assert np.array_equal(np.fromfile(oa, dtype=np.uint8).reshape(1356, 2040, 3), wants[0])
    assert np.array_equal(np.fromfile(ob, dtype=np.uint8).reshape(1356, 2040, 3), wants[1])


def test_generic_runner_from_c(tmp_path):
    """include/sr.h compiled as C11 and driven from C: an RCAN x2 file on a frame of its own size."""
    meta = dict(arch="rcan", n_resgroups=1, n_resblocks=1, scale=2, seed=74, input="range")
    model, _ = H.build_model(meta, "f16")
    wpath, ipath, opath = (str(tmp_path / f) for f in ("rcan_x2.srmi", "in.u8", "out.u8"))
    export.save_weights(model, wpath)
    img = synth.smooth_image(45, 70, 10)
    img.tofile(ipath)
    assert _run([_program("runner_demo"), "sr", wpath, ipath, "70", "45", opath]).startswith("ok 140x90")
    got = np.fromfile(opath, dtype=np.uint8).reshape(90, 140, 3)
    assert np.array_equal(got, model.forward_u8(img, _lib.ROUND_HALF_UP).numpy())


def test_edsr_runner_from_c(tmp_path):
    meta = dict(arch="edsr", n_resblocks=1, n_feats=64, scale=4, seed=73, input="range")
    model, sd = H.build_model(meta, "f16")
    wpath, ipath, opath = (str(tmp_path / f) for f in ("edsr.srmi", "in.u8", "out.u8"))
    export.save_weights(model, wpath)
    img = synth.smooth_image(256, 256, 9)
    img.tofile(ipath)
    _run([_program("runner_demo"), "edsr", wpath, ipath, "256", "256", opath])
    got = np.fromfile(opath, dtype=np.uint8).reshape(1024, 1024, 3)
    assert np.array_equal(got, model.forward_u8(img, _lib.ROUND_HALF_UP).numpy())
    with torch.no_grad():
        ref = O.post_range(H.oracle_forward(meta, sd, O.pre_raw(img)), O.ROUND_HALF_UP)[0]
    assert np.abs(got.astype(int) - ref.astype(int)).max() <= 1


def test_host_buffer_path_matches_device_path():
    """srmi_model_forward_u8_host (pinned staging, copies overlapped with compute across lanes) returns the
    bytes of srmi_model_forward_u8 for every frame, for batches smaller and larger than the lane count."""
    import ctypes as C
    L = _lib.lib()
    meta = dict(arch="rrdbnet", num_block=2, scale=4, seed=74, input="unit")
    model, _ = H.build_model(meta, "f16")
    eng = model.engine()
    for n, h, w in ((1, 40, 56), (7, 33, 47), (2, 64, 48)):
        imgs = synth.random_image(h, w, 11, n=n)
        want = eng.forward_u8(torch.from_numpy(imgs).cuda(), _lib.ROUND_HALF_UP).cpu().numpy()
        got = np.full((n, h * 4, w * 4, 3), 9, dtype=np.uint8)
        rc = L.srmi_model_forward_u8_host(eng._h, imgs.ctypes.data_as(C.c_void_p), got.ctypes.data_as(C.c_void_p), n, h, w,
                                          _lib.ROUND_HALF_UP)
        assert rc == 0, L.srmi_last_error().decode()
        assert np.array_equal(got, want)
