"""Repeats the bench step many times and checks that every repetition reproduces the first output bit for bit
(catches races between the engine's stream lanes and workspace reuse, and in RCAN between the workgroups that
add to the pooling sums and the one that finishes the attention).

    python tools/soak.py [repetitions] [rrdbnet | rcan | rcan_f32trunk | edsr]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from superresolution_amd import EDSR, RCAN, RRDBNet, _lib, synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
which = sys.argv[2] if len(sys.argv) > 2 else "rrdbnet"
h, w = 339, 510
if which == "rrdbnet":
    m, sd = RRDBNet(compute_dtype="f16", trunk_f32=False), synth.rrdbnet_state_dict(seed=0)
elif which.startswith("rcan"):
    m, sd = RCAN(scale=4, compute_dtype="f16", trunk_f32=which == "rcan_f32trunk"), synth.rcan_state_dict(seed=0)
else:
    m, sd = EDSR(compute_dtype="f16", trunk_f32=False), synth.edsr_state_dict(seed=0)
    h, w = 256, 256
m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
eng = m.engine()
imgs = torch.from_numpy(synth.random_image(h, w, 1, n=12)).cuda()
ref = eng.forward_u8(imgs, _lib.ROUND_HALF_UP).clone()
eng.set_lanes(1)
one = eng.forward_u8(imgs, _lib.ROUND_HALF_UP)
assert torch.equal(ref, one), "three lanes and one lane disagree"
eng.set_lanes(3)
out = torch.empty_like(ref)
t0 = time.time()
bad = 0
for i in range(reps):
    eng.forward_u8(imgs, _lib.ROUND_HALF_UP, out=out)
    if not torch.equal(out, ref):
        bad += 1
        print("repetition", i, "differs:", int((out != ref).sum()), "bytes")
    if i % 25 == 24:
        print(f"{i + 1} repetitions, {bad} mismatches, {time.time() - t0:.1f} s", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
