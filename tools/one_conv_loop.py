"""One conv shape launched repeatedly (for PMC passes and kernel traces): one_conv_loop.py cin cout h w mb iters"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_conv import bench
cin, cout, h, w, mb, iters = (int(a) for a in sys.argv[1:7])
us, tf = bench(cin, cout, h, w, mb, iters=iters)
print(f"cin {cin} cout {cout} {w}x{h} mb{mb}: {us:.1f} us {tf:.0f} TF")
