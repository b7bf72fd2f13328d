import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
from superresolution_amd import _lib, ops, synth

def rnd(key, shape, seed=0, amp=1.0):
    return torch.from_numpy(synth.hash_uniform(key, shape, seed)) * amp

for dt in (_lib.F32, _lib.F16):
    print("== dtype", dt)
    n,h,w = 1, 8, 32
    x = rnd("x",(n,64,h,w),1)
    x = x.to(torch.float16).float()
    p = ops.Planes.from_nchw(x, dt)
    rt = p.to_nchw().cpu()
    print("roundtrip max err", (rt-x).abs().max().item())
    # identity: center tap, out ch o = in ch o
    wt = torch.zeros(64,64,3,3); 
    for o in range(64): wt[o,o,1,1]=1.0
    b = torch.zeros(64)
    out = ops.conv3x3(p, ops.PackedConv(wt.numpy(), b.numpy(), dt), out_dtype=_lib.F32).to_nchw().cpu()
    print("identity max err", (out-x).abs().max().item())
    if (out-x).abs().max() > 1e-3:
        # diagnose: which output (c,y,x) equals which input?
        for (c,y,xx) in [(0,0,0),(1,0,0),(0,0,1),(0,1,0),(5,2,3),(33,4,17),(63,7,31)]:
            v = out[0,c,y,xx].item()
            idx = (x[0]-v).abs().argmin().item()
            ci, r = divmod(idx, h*w); yi, xi = divmod(r, w)
            print(f" out[{c},{y},{xx}]={v:.4f}  ~ in[{ci},{yi},{xi}] (in[{c},{y},{xx}]={x[0,c,y,xx]:.4f})")
    # shift: tap (0,0) -> out[y][x] = in[y-1][x-1]
    wt = torch.zeros(64,64,3,3)
    for o in range(64): wt[o,o,0,0]=1.0
    out = ops.conv3x3(p, ops.PackedConv(wt.numpy(), b.numpy(), dt), out_dtype=_lib.F32).to_nchw().cpu()
    ref = F.conv2d(x, wt, b, padding=1)
    print("shift(0,0) max err", (out-ref).abs().max().item())
    # channel mixing: out o = in (o+1)%64 center tap
    wt = torch.zeros(64,64,3,3)
    for o in range(64): wt[o,(o+1)%64,1,1]=1.0
    out = ops.conv3x3(p, ops.PackedConv(wt.numpy(), b.numpy(), dt), out_dtype=_lib.F32).to_nchw().cpu()
    ref = F.conv2d(x, wt, b, padding=1)
    print("chanrot max err", (out-ref).abs().max().item())
    wt = rnd("w",(64,64,3,3),2,0.1).to(torch.float16).float()
    out = ops.conv3x3(p, ops.PackedConv(wt.numpy(), b.numpy(), dt), out_dtype=_lib.F32).to_nchw().cpu()
    ref = F.conv2d(x, wt, b, padding=1)
    print("random max err", (out-ref).abs().max().item(), "scale", ref.abs().max().item())
