"""Per-layer timing of the dense-block convolutions at 510x339, direct form (12-row tiles) against the Winograd form
(HIP-event timing, many launches per shape): python tools/bench_wino.py [H W]."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from superresolution_amd import _lib, ops, synth


def bench(cin, cout, h, w, winograd, iters=60):
    x = ops.Planes(1, cin, h, w, _lib.F16)
    x.buf.normal_()
    wt = synth.hash_uniform("w", (cout, cin, 3, 3), 1) * 0.05
    pc = ops.PackedConv(wt, np.zeros(cout, np.float32), _lib.F16, cout_tile=32, winograd=winograd)
    out = ops.Planes(1, cout, h, w, _lib.F16)
    for _ in range(5):
        ops.conv3x3(x, pc, out=out, act=_lib.ACT_LRELU)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.conv3x3(x, pc, out=out, act=_lib.ACT_LRELU)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    return us, 2.0 * cin * cout * 9 * h * w / us / 1e6


if __name__ == "__main__":
    h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (339, 510)
    print(f"frame {w}x{h}, one launch at a time (one stream)")
    tot = [0.0, 0.0, 0.0]
    for cin, cout in ((64, 32), (96, 32), (128, 32), (160, 32), (192, 64)):
        a, b, c = bench(cin, cout, h, w, 0), bench(cin, cout, h, w, 1), bench(cin, cout, h, w, 2)
        tot[0] += a[0]
        tot[1] += b[0]
        tot[2] += c[0]
        print(f"cin {cin:3d} cout {cout:2d} | direct {a[0]:6.1f} us {a[1]:5.0f} TF | F(2x2,3x3) {b[0]:6.1f} us {b[1]:5.0f} TF (algorithmic) x{a[0] / b[0]:.2f} | "
              f"F(2,3)x {c[0]:6.1f} us {c[1]:5.0f} TF (algorithmic) x{a[0] / c[0]:.2f}")
    print(f"dense block: direct {tot[0]:.1f} us, F(2x2,3x3) {tot[1]:.1f} us (x{tot[0] / tot[1]:.2f}), F(2,3) along x {tot[2]:.1f} us (x{tot[0] / tot[2]:.2f})")
