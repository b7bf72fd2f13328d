"""CPU time the engine needs to enqueue one bench step (12 frames x 351 launches) against the GPU time of the step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from superresolution_amd import RRDBNet, _lib, synth
m = RRDBNet(compute_dtype="f16", trunk_f32=False)
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.rrdbnet_state_dict(seed=0).items()}, strict=True)
eng = m.engine()
imgs = torch.from_numpy(synth.random_image(339, 510, 1, n=12)).cuda()
out = torch.empty((12, 1356, 2040, 3), dtype=torch.uint8, device="cuda")
for _ in range(2):
    eng.forward_u8(imgs, _lib.ROUND_HALF_UP, out=out)
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter()
    eng.forward_u8(imgs, _lib.ROUND_HALF_UP, out=out)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"enqueue {1e3 * (t1 - t0):.1f} ms CPU, step complete after {1e3 * (t2 - t0):.1f} ms ({1e6 * (t1 - t0) / (12 * 351):.2f} us per launch)")
