"""Micro-benchmark of srmi_conv3x3 (HIP-event timing, many launches per config)."""
import sys, os, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from superresolution_amd import _lib, ops, synth

def bench(cin, cout, h, w, mb, dt=_lib.F16, n=1, iters=50, **kw):
    x = ops.Planes(n, cin, h, w, dt, zero=False)
    x.buf.normal_()
    wt = synth.hash_uniform("w", (cout, cin, 3, 3), 1) * 0.05
    pc = ops.PackedConv(wt, np.zeros(cout, np.float32), dt)
    out = ops.Planes(n, cout, h, w, dt)
    for _ in range(3):
        ops.conv3x3(x, pc, out=out, rows_per_wave=mb, act=_lib.ACT_LRELU, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.conv3x3(x, pc, out=out, rows_per_wave=mb, act=_lib.ACT_LRELU, **kw)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    fl = 2.0 * cin * cout * 9 * h * w * n
    return us, fl / us / 1e6

if __name__ == "__main__":
    h, w = 339, 510
    if len(sys.argv) > 2: h, w = int(sys.argv[1]), int(sys.argv[2])
    print(f"frame {w}x{h}; library {os.environ.get('SRMI_LIB', 'in-tree')}")
    for cout in (32, 64):
        for cin in (64, 128, 192, 384, 768):
            row = []
            for mb in ((2, 3, 4) if cout == 32 else (2, 3)):
                us, tf = bench(cin, cout, h, w, mb)
                row.append(f"mb{mb}: {us:7.1f} us {tf:6.0f} TF")
            print(f"cin {cin:4d} cout {cout:3d} | " + " | ".join(row))
