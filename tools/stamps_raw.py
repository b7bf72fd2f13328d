"""Diagnostic: raw per-phase stamp differences of one conv launch (-DSRMI_STAMPS [-DSRMI_STAMPS_EPI] build, SRMI_LIB=...).
usage: stamps_raw.py cin cout rows_per_wave nstamps"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
cin, cout, mb, nst = (int(a) for a in sys.argv[1:5])
h, w = 339, 510
from superresolution_amd import _lib
tiles = _lib.lib().srmi_conv3x3_tiles(h, w, mb) * (cout // 32)
buf = torch.zeros(tiles * 4 * 40, dtype=torch.int64, device="cuda")
os.environ["SRMI_STAMP_PTR"] = str(buf.data_ptr())
from tools.bench_conv import bench
us, tf = bench(cin, cout, h, w, mb, iters=3)
torch.cuda.synchronize()
s = buf.cpu().numpy().reshape(tiles, 4, 40)[:, :, :nst].astype(np.float64)
d = np.diff(s, axis=2)
print(f"cin {cin} cout {cout} mb {mb}: {us:.1f} us, WG lifetime p50 {np.median(s[:, :, -1] - s[:, :, 0]):.0f} cycles")
for i in range(nst - 1):
    v = d[:, :, i].ravel()
    print(f"  phase {i:2d}: p10 {np.percentile(v, 10):7.0f} p50 {np.median(v):7.0f} p90 {np.percentile(v, 90):7.0f}")
