"""Error behaviour of the runner C API (<m>_init / _inference / _inference_ex / _release with null, uninitialised and
unreadable arguments) side by side with the reference runners' own entry points, compiled from the reference's sources
(`make -C oracle ref`, build container only: skipped where /root/reference is absent).  Only calls that return before
the reference touches its NPU runtime are made; `<m>_inference(NULL, ...)` is left out because the reference
dereferences the context there (edsr.cpp:203-207).  Both libraries take the same zero-filled context bytes: the
context structs are identical field for field (include/edsr.h, include/realesrgan.h)."""
import json
import os
import subprocess
import sys

import pytest


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference"),
                                reason="the reference sources are only present in the build container")


@pytest.mark.parametrize("m", ["edsr", "rcan", "realesrgan"])
def test_error_paths_return_what_the_reference_returns(m, tmp_path):
    """Both libraries run the same calls in one short-lived process (tests/ref_cpp_check.py): the test runner itself
    never maps the reference-derived library."""
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True, capture_output=True)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "ref_cpp_check.py"), "errors", m, str(tmp_path)],
                       capture_output=True, text=True, timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
    assert r.returncode == 0 and lines, r.stdout[-2000:] + r.stderr[-2000:]
    res = json.loads(lines[-1][len("RESULT "):])
    want, got = res["ref"], res["mine"]
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
