"""RCAN's 64 -> 64 convolution at 510x339 (4 stages): plain, ReLU, and with the per-tile channel sums; both tile widths."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from superresolution_amd import _lib, ops, synth
dt = _lib.F16
h, w = 339, 510
x = ops.Planes(1, 64, h, w, dt); x.buf.normal_()
wt = synth.hash_uniform("w", (64, 64, 3, 3), 1) * 0.05
out = ops.Planes(1, 64, h, w, dt)
def run(pc, **kw):
    for _ in range(3):
        ops.conv3x3(x, pc, out=out, rows_per_wave=3, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        ops.conv3x3(x, pc, out=out, rows_per_wave=3, **kw)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / 200
for tile in (64, 32):
    pc = ops.PackedConv(wt, np.zeros(64, np.float32), dt, cout_tile=tile)
    print(f"cout_tile {tile}: plain {run(pc):.2f} us | relu {run(pc, act=_lib.ACT_RELU):.2f} us | chansum {run(pc, chansum=True):.2f} us")
