"""Single-frame latency of the three networks: srmi_model_forward_u8 launched directly against the same call captured
into a HIP graph and replayed (one frame, one lane; u8 in HBM -> u8 in HBM, synchronised per frame, wall clock), and the
steady rate of a loop of single-frame replays without synchronising in between.  python tools/latency_graph.py"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from superresolution_amd import _lib, synth

L = _lib.lib()
for arch, h, w in (("rrdbnet", 339, 510), ("rcan", 339, 510), ("edsr", 256, 256)):
    model, _ = bench.build_model(arch, "f16", False, 0)
    eng = model.engine()
    s = eng.scale
    img = torch.from_numpy(synth.random_image(h, w, 5, n=1)).cuda()
    out = torch.zeros((1, h * s, w * s, 3), dtype=torch.uint8, device="cuda")
    for _ in range(3):
        eng.forward_u8(img, _lib.ROUND_HALF_UP, out=out)
    torch.cuda.synchronize()
    want = out.clone()

    def timed(fn, iters=30, sync_each=True):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
            if sync_each:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters * 1e3

    direct = timed(lambda: eng.forward_u8(img, _lib.ROUND_HALF_UP, out=out))
    direct_loop = timed(lambda: eng.forward_u8(img, _lib.ROUND_HALF_UP, out=out), sync_each=False)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        rc = L.srmi_model_forward_u8(eng._h, img.data_ptr(), out.data_ptr(), 1, h, w, _lib.ROUND_HALF_UP, stream)
    assert rc == 0, L.srmi_last_error().decode()
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, want)
    replay = timed(g.replay)
    replay_loop = timed(g.replay, sync_each=False)
    print(f"{arch} {w}x{h}: one frame per call, synchronised: direct launches {direct:.2f} ms, graph replay {replay:.2f} ms; "
          f"back to back without synchronising: direct {direct_loop:.2f} ms, graph {replay_loop:.2f} ms per frame", flush=True)
    del eng, model
    torch.cuda.empty_cache()
