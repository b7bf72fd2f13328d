"""The RGB layer (cout 3, fused post-processing) alone, per frame, beside a full-width layer on the same frame.
SRMI_RGB_GATHER=0 in the environment keeps it on the direct kernel (A/B of csrc/conv_rgb.hip)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from superresolution_amd import _lib, ops, synth


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def run(cin, h, w, mode):
    dt = _lib.F16
    x = ops.Planes(1, cin, h, w, dt); x.buf.normal_()
    wt = synth.hash_uniform("w", (3, cin, 3, 3), 1) * 0.05
    pc = ops.PackedConv(wt, np.zeros(3, np.float32), dt)
    tail = dict(mode=mode, lo=0.0, hi=1.0, scale=255.0, round=0)
    us = timed(lambda: ops.conv3x3(x, pc, tail=tail))
    gbs = h * w * cin * 2 / us / 1e3
    print(f"{cin}->3 {w}x{h} {'u8 NHWC' if mode == _lib.TAIL_U8_NHWC else 'f32 NCHW'}: {us:8.1f} us, input read at {gbs:6.0f} GB/s", flush=True)


print("SRMI_RGB_GATHER =", os.environ.get("SRMI_RGB_GATHER", "(unset: gather kernel)"))
for mode in (_lib.TAIL_U8_NHWC, _lib.TAIL_F32_NCHW):
    run(64, 1356, 2040, mode)     # RealESRGAN_x4plus conv_last at 510x339 x4
    run(256, 1024, 1024, mode)    # EDSR x4 tail at 256x256 x4
    run(64, 1024, 1024, mode)     # RCAN / EDSR-baseline tail
