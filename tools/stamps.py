"""Diagnostic: per-workgroup phase timestamps of the conv kernel (needs a -DSRMI_STAMPS build)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
cin, cout, mb = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
h, w = 339, 510
from superresolution_amd import _lib, ops, synth
tiles = _lib.lib().srmi_conv3x3_tiles(h, w, mb) * (cout // (32 if cout <= 32 else 64))
buf = torch.zeros(tiles * 40, dtype=torch.int64, device="cuda")
os.environ["SRMI_STAMP_PTR"] = str(buf.data_ptr())
from tools.bench_conv import bench
us, tf = bench(cin, cout, h, w, mb, iters=3)
torch.cuda.synchronize()
s = buf.cpu().numpy().reshape(tiles, 40)
ncb = 2 * (cin // 32)  # stages
nst = 2 + 2 * ncb + 1
s = s[:, :nst].astype(np.float64)
t0 = s[:, 0].min()
print(f"cin {cin} cout {cout} mb {mb}: {us:.1f} us; {tiles} WGs; stamps/WG {nst}; (s_memtime ticks = 100MHz? check)")
start = s[:, 0] - t0
dur = s[:, -1] - s[:, 0]
print("WG start (ticks): min %.0f p50 %.0f max %.0f" % (start.min(), np.median(start), start.max()))
print("WG duration: min %.0f p50 %.0f max %.0f" % (dur.min(), np.median(dur), dur.max()))
print("kernel span (ticks):", (s[:, -1].max() - t0))
d = np.diff(s, axis=1)
names = ["prologue(load0+commit+barrier)"]
for cb in range(ncb):
    names += [f"stage{cb} compute", f"stage{cb} wait+barrier"]
names += ["epilogue"]
rows = list(enumerate(names))
if len(rows) > 12: rows = rows[:7] + rows[-5:]
for i, nm in rows:
    print(f"  {nm:34s} p50 {np.median(d[:, i]):8.0f}  p10 {np.percentile(d[:, i],10):8.0f} p90 {np.percentile(d[:, i],90):8.0f}")
