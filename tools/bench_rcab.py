"""Back-to-back launch time of the two convolutions of an RCAN block (64 -> 64 at 510x339, f16) in their
variants: plain, with per-tile channel sums, with the attention finished inside, with a gated input."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from superresolution_amd import _lib, ops, synth
from superresolution_amd._lib import check, lib


def main():
    h, w, c, n = 339, 510, 64, 1
    if len(sys.argv) > 2:
        h, w = int(sys.argv[1]), int(sys.argv[2])
    dt = _lib.F16
    x, t, y, s = (ops.Planes(n, c, h, w, dt) for _ in range(4))
    for p in (x, t):
        p.buf.normal_()
    wt = synth.hash_uniform("w", (c, c, 3, 3), 1) * 0.05
    pc = ops.PackedConv(wt, np.zeros(c, np.float32), dt)
    cr = 4
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32).reshape(-1)).cuda()
    w1, b1, w2, b2 = dev(synth.hash_uniform("w1", (cr, c), 2)), dev(np.zeros(cr)), dev(synth.hash_uniform("w2", (c, cr), 3)), dev(np.zeros(c))
    sums = torch.zeros((n, _lib.CA_SLOTS, c), dtype=torch.int64, device="cuda")
    count = torch.zeros((n,), dtype=torch.int32, device="cuda")
    scale = torch.full((n, c), 0.5, dtype=torch.float32, device="cuda")
    tiles = lib().srmi_conv3x3_tiles(h, w, 3)
    chansum = torch.empty((n, tiles, c), dtype=torch.float32, device="cuda")
    stream = ops._stream()

    def desc(mb, **kw):
        d = _lib.ConvDesc()
        d.dtype, d.n, d.h, d.w, d.cin, d.cout = dt, n, h, w, c, c
        d.ps_r, d.cout_tile, d.rows_per_wave, d.alpha, d.beta = 1, pc.cout_tile, mb, 1.0, 1.0
        d.inp, d.out = x.view(), y.view()
        d.wpk, d.bias = pc.w.data_ptr(), pc.b.data_ptr()
        for k, v in kw.items():
            setattr(d, k, v)
        return d

    att = dict(ca_sums=sums.data_ptr(), ca_count=count.data_ptr(), ca_scale=scale.data_ptr(), ca_w1=w1.data_ptr(),
               ca_b1=b1.data_ptr(), ca_w2=w2.data_ptr(), ca_b2=b2.data_ptr(), ca_cr=cr)
    gate = dict(in_gate=t.view(), gate=scale.data_ptr(), gate_alpha=1.0)
    cases = [("plain mb3", desc(3)), ("plain mb2", desc(2)),
             ("tile sums mb3", desc(3, chansum=chansum.data_ptr())),
             ("attention inside mb3", desc(3, **att)), ("attention inside mb2", desc(2, **att)),
             ("gated mb2", desc(2, **gate)), ("gated mb2 + sum store", desc(2, in_sum=s.view(), **gate)),
             ("gated mb3 + sum store", desc(3, in_sum=s.view(), **gate))]
    print(f"frame {w}x{h}, {c}->{c}; library {os.environ.get('SRMI_LIB', 'in-tree')}")
    for name, d in cases:
        for _ in range(5):
            check(lib().srmi_conv3x3(C.byref(d), stream), name)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        iters = 200
        for _ in range(iters):
            check(lib().srmi_conv3x3(C.byref(d), stream), name)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        print(f"{name:26s} {us:7.2f} us  {2.0 * c * c * 9 * h * w * n / us / 1e6:6.0f} TFLOP/s")


if __name__ == "__main__":
    main()
