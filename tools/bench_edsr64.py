"""EDSR-baseline (16 blocks x 64 features, x4) at 256x256, batch 36: one-kernel ResBlocks against two launches per block, with the
trunk in f16 and in f32 (HIP-event time over 10 calls).  python tools/bench_edsr64.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from superresolution_amd import EDSR, _lib, synth
for trunk in (False, True):
    m = EDSR(n_resblocks=16, n_feats=64, scale=4, compute_dtype="f16", trunk_f32=trunk)
    sd = synth.edsr_state_dict(n_resblocks=16, n_feats=64, scale=4, seed=3)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    eng = m.engine()
    imgs = torch.from_numpy(synth.random_image(256, 256, 5, n=36)).cuda()
    out = torch.empty((36, 1024, 1024, 3), dtype=torch.uint8, device="cuda")
    for on in (True, False, True, False):
        eng.set_fused_rcab(on)
        for _ in range(3): eng.forward_u8(imgs, _lib.ROUND_HALF_UP, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): eng.forward_u8(imgs, _lib.ROUND_HALF_UP, out=out)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 360
        print(f"edsr 16x64 x4 256x256 batch 36 trunk_f32={trunk} one-kernel={on}: {ms:.3f} ms/frame, {1024*1024/ms/1e3:.1f} Mpix/s, {eng.flops(256,256)/ms/1e9:.0f} TF", flush=True)
