"""Path fuzzer: every way of running a batch through one model must give the same bytes.  For random networks,
batch sizes (1-7) and image sizes: srmi_model_forward_u8 with 1-3 lanes and 1-3 frames per launch, the host-buffer
path (srmi_model_forward_u8_host), and the frames run one at a time, in all three rounding modes by turns.
Exit status 1 on the first difference.

    python tools/fuzz_paths.py [cases] [seed]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from superresolution_amd import _lib, synth  # noqa: E402
from tests import helpers as H  # noqa: E402

ARCHS = [
    dict(arch="rrdbnet", num_block=1, scale=4, input="unit", seed=11),
    dict(arch="edsr", n_resblocks=2, n_feats=64, scale=2, input="raw", seed=12),
    dict(arch="rcan", scale=3, n_resgroups=1, n_resblocks=2, input="raw", seed=13),
]


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    engines = [(H.build_model(m, "f16")[0]) for m in ARCHS]
    bad = 0
    for i in range(cases):
        model = engines[i % len(ARCHS)]
        eng = model.engine()
        n, h, w = int(rng.integers(1, 8)), int(rng.integers(1, 70)), int(rng.integers(1, 100))
        mode = [_lib.ROUND_HALF_UP, _lib.ROUND_HALF_EVEN, _lib.ROUND_TRUNC][i % 3]
        frames = synth.random_image(h, w, 7000 + i, n=n)
        dev = torch.from_numpy(frames).cuda()
        eng.set_lanes(1)
        eng.set_sweep(1)
        want = torch.cat([eng.forward_u8(dev[j:j + 1], mode) for j in range(n)]).cpu().numpy()
        got = {}
        for lanes, sweep in ((3, 1), (2, 2), (1, 3), (3, 3)):
            eng.set_lanes(lanes)
            eng.set_sweep(sweep)
            got[f"lanes {lanes} sweep {sweep}"] = eng.forward_u8(dev, mode).cpu().numpy()
        eng.set_lanes(3)
        eng.set_sweep(1)
        got["host path"] = eng.forward_u8_host(frames, mode)
        for k, v in got.items():
            if v.shape != want.shape or not np.array_equal(v, want):
                bad += 1
                print(f"FAIL {ARCHS[i % len(ARCHS)]['arch']} n={n} h={h} w={w} mode={mode}: {k} differs from single frames", flush=True)
        if (i + 1) % 10 == 0:
            print(f"{i + 1}/{cases} cases, {bad} differences", flush=True)
    print(f"fuzz_paths: {cases} cases, seed {seed}, {bad} differences")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
