"""Error behaviour of the runner C API (<m>_init / _inference / _inference_ex / _release with null, uninitialised and
unreadable arguments) side by side with the reference runners' own entry points, compiled from the reference's sources
(`make -C oracle ref`, build container only: skipped where /root/reference is absent).  Only calls that return before
the reference touches its NPU runtime are made; `<m>_inference(NULL, ...)` is left out because the reference
dereferences the context there (edsr.cpp:203-207).  Both libraries take the same zero-filled context bytes: the
context structs are identical field for field (include/edsr.h, include/realesrgan.h)."""
import ctypes as C
import os
import subprocess

import pytest

from superresolution_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference"),
                                reason="the reference sources are only present in the build container")


def _bind(L, m):
    f = {}
    for name, args in (("init", [C.c_char_p, C.c_void_p]), ("release", [C.c_void_p]),
                       ("inference", [C.c_void_p, C.c_void_p, C.c_void_p]),
                       ("inference_ex", [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p])):
        fn = getattr(L, f"{m}_{name}")
        fn.argtypes, fn.restype = args, C.c_int
        f[name] = fn
    return f


def _calls(f, tmp_path):
    """Return codes of the safe error-path calls, in a fixed order."""
    ctx = C.create_string_buffer(512)           # zero-filled: an uninitialised context
    img = C.create_string_buffer(4 * 4 * 3)
    out = C.c_void_p(0)
    ow, oh = C.c_int(0), C.c_int(0)
    garbage = tmp_path / "garbage.bin"
    garbage.write_bytes(b"not a model" * 10)
    rc = {}
    rc["init(NULL, ctx)"] = f["init"](None, ctx)
    rc["init(path, NULL)"] = f["init"](b"/nonexistent/model.bin", None)
    rc["init(missing file, ctx)"] = f["init"](b"/nonexistent/model.bin", ctx)
    rc["ctx still zero after failed init"] = int(ctx.raw == bytes(512))
    rc["init(garbage file, ctx)"] = f["init"](str(garbage).encode(), ctx)
    rc["release(NULL)"] = f["release"](None)
    ctx = C.create_string_buffer(512)
    rc["release(uninitialised ctx)"] = f["release"](ctx)
    rc["inference(uninitialised ctx)"] = f["inference"](ctx, img, C.byref(out))
    rc["inference_ex(NULL ctx)"] = f["inference_ex"](None, img, 4, 4, C.byref(out), C.byref(ow), C.byref(oh))
    rc["inference_ex(uninitialised ctx)"] = f["inference_ex"](ctx, img, 4, 4, C.byref(out), C.byref(ow), C.byref(oh))
    rc["inference_ex(uninitialised ctx, NULL sizes)"] = f["inference_ex"](ctx, img, 4, 4, C.byref(out), None, None)
    rc["output pointer untouched"] = int(not out.value)
    return rc


@pytest.mark.parametrize("m", ["edsr", "rcan", "realesrgan"])
def test_error_paths_return_what_the_reference_returns(m, tmp_path):
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True, capture_output=True)
    ref = _bind(C.CDLL(os.path.join(ROOT, "oracle", "_ref", f"libref_{m}.so")), m)
    mine = _bind(_lib.lib(), m)
    want, got = _calls(ref, tmp_path), _calls(mine, tmp_path)
    print(m, want)
    assert got == want
    # the documented difference between the runners (edsr.cpp:185-187 against realesrgan.cpp:197-199)
    assert want["release(uninitialised ctx)"] == (-1 if m == "realesrgan" else 0)


def test_executor_class_behaves_like_the_reference_class_before_initialisation():
    """Mi355Executor against the reference's NeuronExecutor (neuron_executor.h:85-117, its own .cpp compiled where it
    lies): tensor counts and byte sizes from the constructor's shapes, Run() before Initialize(), Initialize() on a
    missing model file (oracle/ref_executor_compare.cpp)."""
    ref_src = "/root/reference/edsr/mtk/cpp/jni/src"
    out = os.path.join(ROOT, "oracle", "_ref", "executor_compare")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    lib_dir = os.path.join(ROOT, "superresolution_amd")
    subprocess.run(["g++", "-std=c++17", "-O1", "-w", "-I" + os.path.join(ROOT, "include"), "-I" + ref_src,
                    os.path.join(ROOT, "oracle", "ref_executor_compare.cpp"), os.path.join(ref_src, "mtk_npu", "neuron_executor.cpp"),
                    "-L" + lib_dir, "-lsrmi355", "-Wl,-rpath," + lib_dir, "-Wl,-rpath-link,/opt/rocm/lib", "-L/opt/rocm/lib",
                    "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib", "-ldl", "-o", out], check=True)
    r = subprocess.run([out], capture_output=True, text=True, timeout=120)
    print(r.stdout[-2000:], r.stderr[-1000:])
    assert r.returncode == 0 and "executor semantics match" in r.stdout
