"""CPU emulation of EDSR's rounding schedules on the oracle (tools only): which layers may use bf16 operands while the u8
output stays within 1 LSB / 50 dB of the fp32 oracle.  The engine's schedule: conv operands (activations as stored,
weights) rounded to the layer's type, f32 accumulation, f32 residual trunk, results stored in the type of the layer that
reads them.  python tools/edsr_mixed_emulation.py [size]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import torch.nn.functional as F

import helpers as H
from oracle import sr_oracle as O
from superresolution_amd import synth

torch.set_num_threads(min(16, os.cpu_count() or 1))
size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
meta = dict(arch="edsr", n_resblocks=32, n_feats=256, scale=4, seed=32, input="range")
sd = H.state_dict_for(meta)
r16 = lambda t: t.to(torch.float16).to(torch.float32)
rb = lambda t: t.to(torch.bfloat16).to(torch.float32)
sat16 = lambda t: r16(t.clamp(-65504.0, 65504.0))


def conv(name, x, rnd):
    return F.conv2d(rnd(x), rnd(O._t(sd, name + ".weight")), O._t(sd, name + ".bias"), padding=1)


def forward(x, body, last, tail):
    """body / last / tail: rounding of the operands of the residual blocks, of body.32 and of the upsampler + tail."""
    x = O.meanshift(sd, "sub_mean", x)
    head = F.conv2d(x, O._t(sd, "head.0.weight"), O._t(sd, "head.0.bias"), padding=1)   # first layer computes in f32
    res = head
    for i in range(32):
        t = F.relu(conv(f"body.{i}.body.0", res, body))
        res = conv(f"body.{i}.body.2", t, body) * 0.1 + res          # f32 trunk
    res = conv("body.32", res, last) + head
    y = O.pixel_shuffle(conv("tail.0.0", res, tail), 2)
    y = O.pixel_shuffle(conv("tail.0.2", y, tail), 2)
    y = conv("tail.1", y, tail)
    return O.meanshift(sd, "add_mean", y)


imgs = {"smooth": synth.smooth_image(size, size, 40), "noise": synth.random_image(size, size, 41)}
print(f"| frame {size}x{size} | body.0-31 | body.32 | tail | max u8 diff | > 1 LSB | PSNR vs fp32 oracle |")
print("|---|---|---|---|---|---|---|")
with torch.no_grad():
    for kind, img in imgs.items():
        x = O.pre_raw(img)
        ref = O.post_range(O.edsr_forward(sd, x), O.ROUND_HALF_UP)[0]
        for names, fns in ((("f16", "f16", "f16"), (r16, r16, r16)), (("bf16", "bf16", "bf16"), (rb, rb, rb)),
                           (("bf16", "bf16", "f16"), (rb, rb, r16)), (("bf16", "f16", "f16"), (rb, r16, r16))):
            got = O.post_range(forward(x, *fns), O.ROUND_HALF_UP)[0]
            d = np.abs(got.astype(np.int32) - ref.astype(np.int32))
            print(f"| {kind} | {names[0]} | {names[1]} | {names[2]} | {d.max()} | {100 * (d > 1).mean():.4f} % | {O.psnr_u8(got, ref):.2f} dB |", flush=True)
