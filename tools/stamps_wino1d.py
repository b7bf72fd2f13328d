"""Diagnostic: per-wave phase timestamps of the F(2,3)-along-x convolution kernel (needs a -DSRMI_STAMPS build, SRMI_LIB=...):
tools/build_variant.sh stamps -DSRMI_STAMPS && SRMI_LIB=.../libsrmi355_stamps.so python tools/stamps_wino1d.py [cin cout h w]"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from superresolution_amd import _lib, ops, synth

cin, cout, h, w = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (128, 32, 339, 510)
tiles = ((h + 7) // 8) * ((w + 63) // 64) * (cout // 32)
buf = torch.zeros(tiles * 4 * 40, dtype=torch.int64, device="cuda")
os.environ["SRMI_STAMP_PTR"] = str(buf.data_ptr())
X = ops.Planes(1, cin, h, w, _lib.F16)
X.buf.normal_()
Y = ops.Planes(1, cout, h, w, _lib.F16)
pc = ops.PackedConv(synth.hash_uniform("w", (cout, cin, 3, 3), 1) * 0.05, np.zeros(cout, np.float32), _lib.F16, winograd=2)
for _ in range(3):
    ops.conv3x3(X, pc, act=_lib.ACT_LRELU, out=Y)
torch.cuda.synchronize()
raw = buf.cpu().numpy().reshape(tiles, 4, 40)
nstage = cin // 16
names = ["prologue issue"]
for k in range(nstage):
    names += [f"st{k} wait+barrier", f"st{k} compute"]
names += ["closing barriers", "epilogue (stores drained)"]
s = raw[:, :, :len(names) + 1].astype(np.float64)
hw = raw[:, :, 39]
k = 1e-3
life = (s[:, 0, -1] - s[:, 0, 0]) * k
print(f"cin {cin} cout {cout} {w}x{h}: {tiles} workgroups x 4 waves; workgroup life p50 {np.median(life):.1f} kcyc (MFMA issue alone: {24 * 32 * nstage * k:.1f})")
d = np.diff(s, axis=2) * k
show = set(range(0, 5)) | set(range(len(names) - 4, len(names)))
for i, nm in enumerate(names):
    if i not in show:
        continue
    v = d[:, :, i].ravel()
    print(f"  {nm:28s} p10 {np.percentile(v, 10):6.2f} p50 {np.median(v):6.2f} p90 {np.percentile(v, 90):6.2f} mean {v.mean():6.2f} kcyc")
comp = d[:, :, 2:2 + 2 * nstage:2]
wait = d[:, :, 1:1 + 2 * nstage:2]
print(f"per stage, stages >= 1: compute mean {comp[:, :, 1:].mean():.2f}, wait+barrier mean {wait[:, :, 1:].mean():.2f} kcyc (MFMA issue alone 0.77)")
groups = collections.defaultdict(list)
for t in range(tiles):
    groups[(int(hw[t, 0] >> 32) & 0xF, (int(hw[t, 0]) >> 8) & 0xFF)].append(t)
print("CUs used:", len(groups), "workgroups per CU histogram:", dict(collections.Counter(len(v) for v in groups.values())))
