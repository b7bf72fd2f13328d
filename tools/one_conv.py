"""Run one conv config a few times (for rocprofv3 counter passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_conv import bench
cin, cout, mb = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
h, w = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (339, 510)
us, tf = bench(cin, cout, h, w, mb, iters=5)
print(f"cin {cin} cout {cout} mb {mb} {w}x{h}: {us:.1f} us {tf:.0f} TF")
