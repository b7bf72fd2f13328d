"""One RCAB body at 510x339: the one-kernel form against the two-launch form (HIP-event timing):
python tools/bench_rcab_fused.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from superresolution_amd import _lib, ops, synth

h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (339, 510)
c, cr, iters = 64, 4, 60
X, Y = ops.Planes(1, c, h, w, _lib.F16), ops.Planes(1, c, h, w, _lib.F16)
X.buf.normal_()
Y.buf.normal_()
g = torch.rand((1, c), device="cuda")
mk = lambda s: ops.PackedConv(synth.hash_uniform("w", (c, c, 3, 3), s) * 0.05, np.zeros(c, np.float32), _lib.F16, cout_tile=64)
pc1, pc2 = mk(1), mk(2)
att = (synth.hash_uniform("a", (cr, c), 3), np.zeros(cr, np.float32), synth.hash_uniform("b", (c, cr), 4), np.zeros(c, np.float32))
S = ops.Planes(1, c, h, w, _lib.F16)


def timed(fn):
    """Device time per call: the calls are captured into a HIP graph and replayed, so the Python binding's host time per
    launch (tens of microseconds) is not what is measured."""
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


T = ops.Planes(1, c, h, w, _lib.F16)
O2 = ops.Planes(1, c, h, w, _lib.F16)
# (the attention variants synchronise per call in the Python wrapper: timed here without the attention tail, which costs
# the same atomics in both forms)
two = timed(lambda: (ops.conv3x3(X, pc1, act=_lib.ACT_RELU, rows_per_wave=2, in_gate=Y, gate=g, gate_alpha=1.0, in_sum=S, out=T),
                     ops.conv3x3(T, pc2, rows_per_wave=3, out=O2)))
O1 = ops.Planes(1, c, h, w, _lib.F16)
one = timed(lambda: ops.rcab_fused(X, pc1, pc2, in_gate=Y, gate=g, gate_alpha=1.0, in_sum=S, out=O1))
fl = 2 * 2.0 * c * c * 9 * h * w
print(f"{w}x{h} ({((h + 7) // 8) * ((w + 31) // 32)} workgroups of the one-kernel form): two launches {two:.1f} us ({fl / two / 1e6:.0f} TF), one kernel {one:.1f} us ({fl / one / 1e6:.0f} TF algorithmic), x{two / one:.2f}")
