"""Tile variants on the x4-resolution layers (64 -> 64 at 2040x1356, and the nearest-x2 loader at 1020x678 -> 2040x1356)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from superresolution_amd import _lib, ops, synth

def run(cin, cout, h, w, mb, tile, up2=False, iters=10):
    dt = _lib.F16
    x = ops.Planes(1, cin, h // 2 if up2 else h, w // 2 if up2 else w, dt); x.buf.normal_()
    wt = synth.hash_uniform("w", (cout, cin, 3, 3), 1) * 0.05
    pc = ops.PackedConv(wt, np.zeros(cout, np.float32), dt, cout_tile=tile)
    out = ops.Planes(1, cout, h, w, dt)
    for _ in range(2):
        ops.conv3x3(x, pc, out=out, rows_per_wave=mb, act=_lib.ACT_LRELU, up2=up2)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.conv3x3(x, pc, out=out, rows_per_wave=mb, act=_lib.ACT_LRELU, up2=up2)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    return us, 2.0 * cin * cout * 9 * h * w / us / 1e6

for name, (h, w, up2) in {"conv_hr 2040x1356": (1356, 2040, False), "conv_up2 (nearest x2 loader)": (1356, 2040, True),
                          "conv_up1 1020x678 (x2 loader)": (678, 1020, True)}.items():
    for tile, mbs in ((32, (2, 3, 4)), (64, (2, 3))):
        print(name, f"cout_tile {tile}:", " | ".join("mb%d %7.1f us %4.0f TF" % ((mb,) + run(64, 64, h, w, mb, tile, up2)) for mb in mbs))
