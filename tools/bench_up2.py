"""The upsampling convolutions of RRDBNet (64 -> 64 behind a nearest-x2 upsampling) alone: nine taps through the x2 address
map against the sub-pixel form (four 2x2 convolutions on the source grid), per tile shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from superresolution_amd import _lib, ops, synth


def run(h, w, mb, tile, sub, iters=20):
    dt = _lib.F16
    x = ops.Planes(1, 64, h // 2, w // 2, dt); x.buf.normal_()
    wt = synth.hash_uniform("w", (64, 64, 3, 3), 1) * 0.05
    pc = ops.PackedConv(wt, np.zeros(64, np.float32), dt, cout_tile=tile, subpixel_up2=sub)
    out = ops.Planes(1, 64, h, w, dt)
    f = lambda: ops.conv3x3(x, pc, out=out, rows_per_wave=mb, act=_lib.ACT_LRELU, up2=True)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for name, (h, w) in {"conv_up2 1020x678 -> 2040x1356": (1356, 2040), "conv_up1 510x339 -> 1020x678": (678, 1020)}.items():
    for sub in (False, True):
        for tile in (32, 64):
            print(name, "sub-pixel" if sub else "nine taps", f"cout_tile {tile}:",
                  " | ".join("mb%d %7.1f us" % (mb, run(h, w, mb, tile, sub)) for mb in (2, 3)), flush=True)
