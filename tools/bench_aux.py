"""HBM-bound kernels at the bench sizes (510x339 by default, 64 channels, f16).  Run under rocprofv3 --kernel-trace --stats;
`--report <kernel_stats.csv>` joins the measured average durations with the algorithmic bytes below and prints the
achieved GB/s against the 8 TB/s HBM peak."""
import csv, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# frame size: SRMI_AUX_SIZE=WxH in the environment (default the bench frame 510x339; 1920x1080 is the large-frame case)
W, H = (int(v) for v in os.environ.get("SRMI_AUX_SIZE", "510x339").split("x"))
C = 64
PX = H * W
# algorithmic bytes per launch (each tensor read or written once)
BYTES = {
    "conv_first_kernel": PX * 3 + PX * C * 2,
    "ca_apply_kernel": 3 * PX * C * 2,
    "ca_scale_kernel": ((H + 11) // 12) * ((W + 31) // 32) * C * 4,
    "channel_sums_kernel": PX * C * 2,
    "pixel_shuffle": 2 * PX * 4 * C * 2,
    "nearest_up2_kernel": PX * C * 2 + 4 * PX * C * 2,
    "nchw_to_planes_kernel": PX * C * 4 + PX * C * 2,
    "planes_to_nchw_kernel": PX * C * 2 + PX * C * 4,
    "resize_bilinear_u8_kernel": 1080 * 1920 * 3 + PX * 3,
}

if len(sys.argv) > 2 and sys.argv[1] == "--report":
    rows = list(csv.DictReader(open(sys.argv[2])))
    print("| kernel | algorithmic bytes / launch | avg launch (us) | achieved GB/s | of 8 TB/s |")
    print("|---|---|---|---|---|")
    for key, b in BYTES.items():
        for r in rows:
            if key in r["Name"]:
                us = float(r["AverageNs"]) / 1e3
                gbs = b / us / 1e3
                print(f"| `{key}` | {b / 1e6:.2f} MB | {us:.2f} | {gbs:.0f} | {gbs / 8000:.3f} |")
    sys.exit(0)

import numpy as np, torch
from superresolution_amd import _lib, ops, synth
dt = _lib.F16
img = torch.from_numpy(synth.random_image(H, W, 1, n=1)).cuda()
w0 = synth.hash_uniform("w0", (C, 3, 3, 3), 1) * 0.1
b0 = np.zeros(C, np.float32)
x = ops.Planes(1, C, H, W, dt); x.buf.normal_()
y = ops.Planes(1, C, H, W, dt); y.buf.normal_()
x4 = ops.Planes(1, 4 * C, H, W, dt); x4.buf.normal_()
TILES = ((H + 11) // 12) * ((W + 31) // 32)
sums = torch.randn(1, TILES, C, device="cuda")
w1, b1 = synth.hash_uniform("w1", (4, C), 2), np.zeros(4, np.float32)
w2, b2 = synth.hash_uniform("w2", (C, 4), 3), np.zeros(C, np.float32)
scale = torch.rand(1, C, device="cuda")
nchw = torch.randn(1, C, H, W, device="cuda")
big = torch.from_numpy(synth.random_image(1080, 1920, 2)).cuda()
for _ in range(20):
    ops.conv_first(img, w0, b0, dt)
    ops.ca_apply(y, x, scale)
    ops.ca_scale(sums, PX, w1, b1, w2, b2)
    ops.channel_sums(x)
    ops.pixel_shuffle(x4, 2)
    ops.nearest_up2(x)
    p = ops.Planes.from_nchw(nchw, dt)
    p.to_nchw()
    ops.resize_bilinear_u8(big, W, H)
torch.cuda.synchronize()
print("done")
