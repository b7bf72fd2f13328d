"""Diagnostic: per-wave phase timestamps of the conv kernel (needs a -DSRMI_STAMPS build, SRMI_LIB=...).
usage: stamps.py cin cout rows_per_wave"""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
cin, cout, mb = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
h, w = 339, 510
from superresolution_amd import _lib, ops, synth
tiles = _lib.lib().srmi_conv3x3_tiles(h, w, mb) * (cout // (32 if cout <= 32 else 64))
buf = torch.zeros(tiles * 4 * 40, dtype=torch.int64, device="cuda")
os.environ["SRMI_STAMP_PTR"] = str(buf.data_ptr())
from tools.bench_conv import bench
us, tf = bench(cin, cout, h, w, mb, iters=3)
torch.cuda.synchronize()
raw = buf.cpu().numpy().reshape(tiles, 4, 40)
nstage = 2 * (cin // 32)
nst = 2 + 2 * nstage + 1
s = raw[:, :, :nst].astype(np.float64)
hw = raw[:, :, 39]
t0 = s[:, :, 0].min()
s -= t0
print(f"cin {cin} cout {cout} mb {mb}: {us:.1f} us; {tiles} WGs x 4 waves; kernel span {s[:, :, -1].max():.0f} cycles")
print("WG start: p50 %.0f max %.0f | WG duration p50 %.0f" % (np.median(s[:, 0, 0]), s[:, 0, 0].max(), np.median(s[:, 0, -1] - s[:, 0, 0])))
d = np.diff(s, axis=2)  # [tile, wave, phase]
names = ["prologue"]
for k in range(nstage):
    names += [f"st{k} compute", f"st{k} wait+barrier"]
names[1] = "prologue wait+barrier"; names = ["prologue issue", "first wait+barrier"] + [x for k in range(nstage) for x in (f"st{k} compute", f"st{k} wait+barrier")]
names[-1] = "epilogue"
# stamps: 0 start, 1 after prologue issue, then (after barrier, after compute) per stage, then after epilogue
names = ["prologue issue"] + [x for k in range(nstage) for x in (f"st{k} wait+barrier", f"st{k} compute")] + ["tail barrier+epilogue"]
for i, nm in enumerate(names):
    if 6 < i < len(names) - 5: continue
    v = d[:, :, i].ravel()
    print(f"  {nm:24s} p10 {np.percentile(v, 10):7.0f} p50 {np.median(v):7.0f} p90 {np.percentile(v, 90):7.0f}")
comp = d[:, :, 2::2][:, :, :nstage]
wait = d[:, :, 1::2][:, :, :nstage]
print(f"per stage (all waves, stages>=1): compute mean {comp[:, :, 1:].mean():.0f}, wait+barrier mean {wait[:, :, 1:].mean():.0f}")
# skew between the four waves of a workgroup at the end of each compute phase
ends = s[:, :, 3::2][:, :, :nstage]
skew = ends.max(axis=1) - ends.min(axis=1)
print(f"compute-end skew inside a workgroup: p50 {np.median(skew):.0f} p90 {np.percentile(skew, 90):.0f}")
# co-residency: group workgroups by CU
cu_key = lambda v: (int(v >> 32) & 0xF, int(v) & 0xFF00 & ~0x00)  # xcc, (se, sh, cu) bits 8..15
groups = collections.defaultdict(list)
for t in range(tiles):
    groups[(int(hw[t, 0] >> 32) & 0xF, (int(hw[t, 0]) >> 8) & 0xFF)].append(t)
sizes = collections.Counter(len(v) for v in groups.values())
print("CUs used:", len(groups), "workgroups per CU histogram:", dict(sizes))
pairs = [v for v in groups.values() if len(v) == 2]
if pairs:
    # overlap of compute intervals of the two workgroups' waves that share a SIMD
    ov_tot, comp_tot = 0.0, 0.0
    for a, b in pairs:
        simd_a = {(int(hw[a, wv]) >> 4) & 3: wv for wv in range(4)}
        simd_b = {(int(hw[b, wv]) >> 4) & 3: wv for wv in range(4)}
        for sd in simd_a:
            if sd not in simd_b: continue
            wa, wb = simd_a[sd], simd_b[sd]
            ia = [(s[a, wa, 2 + 2 * k], s[a, wa, 3 + 2 * k]) for k in range(nstage)]
            ib = [(s[b, wb, 2 + 2 * k], s[b, wb, 3 + 2 * k]) for k in range(nstage)]
            for x0, x1 in ia:
                comp_tot += x1 - x0
                for y0, y1 in ib:
                    ov_tot += max(0.0, min(x1, y1) - max(x0, y0))
    print(f"same-SIMD wave pairs: {100 * ov_tot / comp_tot:.0f} % of a wave's compute time overlaps its partner's compute")
    a, b = pairs[len(pairs) // 2]
    print("example CU, wave 0 of both workgroups (barrier exit -> compute end per stage):")
    for t in (a, b):
        print("   WG", t, "simd of waves", [(int(hw[t, wv]) >> 4) & 3 for wv in range(4)],
              " ".join(f"[{s[t, 0, 2 + 2 * k]:.0f}-{s[t, 0, 3 + 2 * k]:.0f}]" for k in range(min(nstage, 8))))
