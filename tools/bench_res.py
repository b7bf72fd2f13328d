"""Cost of the residual read in the epilogue: RDB conv5 (192 -> 64, 32-wide tiles) with and without residuals."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from superresolution_amd import _lib, ops, synth
dt = _lib.F16
h, w = 339, 510
x = ops.Planes(1, 192, h, w, dt); x.buf.normal_()
r1 = ops.Planes(1, 64, h, w, dt); r1.buf.normal_()
r2 = ops.Planes(1, 64, h, w, dt); r2.buf.normal_()
wt = synth.hash_uniform("w", (64, 192, 3, 3), 1) * 0.05
pc = ops.PackedConv(wt, np.zeros(64, np.float32), dt, cout_tile=32)
out = ops.Planes(1, 64, h, w, dt)
out2 = ops.Planes(1, 64, h, w, dt)
def run(**kw):
    for _ in range(3):
        ops.conv3x3(x, pc, out=out, rows_per_wave=3, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        ops.conv3x3(x, pc, out=out, rows_per_wave=3, **kw)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / 200
for r in range(3):
    print(f"no residual {run():.2f} us | res1 {run(res1=r1, alpha=0.2):.2f} us | "
          f"res1+res2+out2 {run(res1=r1, alpha=0.2, res2=r2, beta=0.2, out2=out2):.2f} us")
