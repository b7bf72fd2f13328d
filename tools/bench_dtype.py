"""f16 against bf16 on the same conv shapes (interleaved, one process), and f16 kernels fed with operands that only
carry 8 significant bits (bf16-representable values): separates the instruction from the data it multiplies."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from superresolution_amd import _lib, ops, synth

def run(cin, cout, dt, quant_x=False, quant_w=False, iters=100):
    h, w = 339, 510
    x = ops.Planes(1, cin, h, w, dt, zero=False)
    x.buf.normal_()
    if quant_x:
        x.buf.copy_(x.buf.to(torch.bfloat16).to(x.buf.dtype))
    wt = synth.hash_uniform("w", (cout, cin, 3, 3), 1) * 0.05
    if quant_w:
        wt = torch.from_numpy(wt).to(torch.bfloat16).to(torch.float32).numpy()
    pc = ops.PackedConv(wt, np.zeros(cout, np.float32), dt)
    out = ops.Planes(1, cout, h, w, dt)
    for _ in range(3):
        ops.conv3x3(x, pc, out=out, rows_per_wave=3, act=_lib.ACT_LRELU)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.conv3x3(x, pc, out=out, rows_per_wave=3, act=_lib.ACT_LRELU)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters

for r in range(3):
    for cin in (192, 768):
        a = run(cin, 32, _lib.F16)
        b = run(cin, 32, _lib.BF16)
        c = run(cin, 32, _lib.F16, quant_x=True)
        d = run(cin, 32, _lib.F16, quant_x=True, quant_w=True)
        print(f"cin {cin:4d}: f16 {a:6.2f} | bf16 {b:6.2f} | f16 kernel, 8-bit activations {c:6.2f} | + 8-bit weights {d:6.2f} us")
