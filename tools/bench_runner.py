"""PCIe-inclusive rate of the runner C API (host u8 frame in -> malloc()ed host u8 frame out), one frame per call."""
import ctypes as C, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from superresolution_amd import RRDBNet, _lib, export, synth
from superresolution_amd.demo import RealesrganContext

m = RRDBNet(compute_dtype="f16", trunk_f32=False)
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.rrdbnet_state_dict(seed=0).items()}, strict=True)
path = os.path.join(tempfile.mkdtemp(), "RealESRGAN_x4plus.srmi")
export.save_weights(m, path)
L = _lib.lib()
ctx = RealesrganContext()
assert L.realesrgan_init(path.encode(), C.byref(ctx)) == 0
libc = C.CDLL(None); libc.free.argtypes = [C.c_void_p]
img = synth.smooth_image(339, 510, 0)
out = C.c_void_p()
for _ in range(3):
    assert L.realesrgan_inference(C.byref(ctx), img.ctypes.data_as(C.c_void_p), C.byref(out)) == 0
    libc.free(out)
n = 20
t0 = time.perf_counter()
for _ in range(n):
    L.realesrgan_inference(C.byref(ctx), img.ctypes.data_as(C.c_void_p), C.byref(out))
    libc.free(out)
dt = (time.perf_counter() - t0) / n
print(f"runner API, host buffers, one 510x339 frame per call: {dt * 1e3:.2f} ms/frame, {2040 * 1356 / dt / 1e6:.1f} output Mpix/s")
L.realesrgan_release(C.byref(ctx))
