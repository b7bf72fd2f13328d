"""Diagnostic: per-wave phase timestamps of the one-kernel RCAN block (needs a -DSRMI_STAMPS build, SRMI_LIB=...):
tools/build_variant.sh stamps -DSRMI_STAMPS && SRMI_LIB=superresolution_amd/variants/libsrmi355_stamps.so python tools/stamps_rcab.py [h w]"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from superresolution_amd import _lib, ops, synth

h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (339, 510)
c = 64
tiles = ((h + 7) // 8) * ((w + 31) // 32)
buf = torch.zeros(tiles * 4 * 40, dtype=torch.int64, device="cuda")
os.environ["SRMI_STAMP_PTR"] = str(buf.data_ptr())
X, Y, S = (ops.Planes(1, c, h, w, _lib.F16) for _ in range(3))
X.buf.normal_()
Y.buf.normal_()
g = torch.rand((1, c), device="cuda")
mk = lambda s: ops.PackedConv(synth.hash_uniform("w", (c, c, 3, 3), s) * 0.05, np.zeros(c, np.float32), _lib.F16, cout_tile=64)
pc1, pc2 = mk(1), mk(2)
for _ in range(3):
    ops.rcab_fused(X, pc1, pc2, in_gate=Y, gate=g, gate_alpha=1.0, in_sum=S)
torch.cuda.synchronize()
raw = buf.cpu().numpy().reshape(tiles, 4, 40)
names = ["prologue issue", "gate vector wait+barrier", "stage-0 x' steps"]
for k in range(4):
    names += [f"conv1 st{k} wait+barrier", f"conv1 st{k} compute"]
names += ["post-conv1 barrier", "conv2 w issue + t epilogue"]
for k in range(4):
    names += [f"conv2 st{k} wait+barrier", f"conv2 st{k} compute"]
names += ["two barriers + bias", "epilogue (stores drained)"]
nst = len(names) + 1
s = raw[:, :, :nst].astype(np.float64)
hw = raw[:, :, 39]
s -= s[:, :, 0].min()
# s_memtime counts shader cycles: print in thousands of cycles (about half a microsecond each at the 2 GHz the chip holds here)
tick_us = 1e-3
print(f"{w}x{h}: {tiles} workgroups x 4 waves; kernel span {s[:, :, -1].max() * tick_us:.1f} kcyc")
print("workgroup start: p50 %.1f max %.1f kcyc | workgroup life p50 %.1f kcyc" % (np.median(s[:, 0, 0]) * tick_us, s[:, 0, 0].max() * tick_us, np.median(s[:, 0, -1] - s[:, 0, 0]) * tick_us))
d = np.diff(s, axis=2) * tick_us
for i, nm in enumerate(names):
    v = d[:, :, i].ravel()
    print(f"  {nm:30s} p10 {np.percentile(v, 10):6.2f} p50 {np.median(v):6.2f} p90 {np.percentile(v, 90):6.2f} mean {v.mean():6.2f} kcyc")
groups = collections.defaultdict(list)
for t in range(tiles):
    groups[(int(hw[t, 0] >> 32) & 0xF, (int(hw[t, 0]) >> 8) & 0xFF)].append(t)
print("CUs used:", len(groups), "workgroups per CU histogram:", dict(collections.Counter(len(v) for v in groups.values())))
first = s[:, 0, 0] < np.percentile(s[:, 0, 0], 50)
print("life of first-round workgroups p50 %.1f kcyc, of later ones %.1f kcyc" % (np.median((s[:, 0, -1] - s[:, 0, 0])[first]) * tick_us, np.median((s[:, 0, -1] - s[:, 0, 0])[~first]) * tick_us))
