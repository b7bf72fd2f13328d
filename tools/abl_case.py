"""Ablation timings (one line): long and typical reductions at 510x339, 32 output channels, 12-row tiles."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_conv import bench
print(" | ".join("cin %4d: %6.1f us %5.0f TF" % ((cin,) + bench(cin, 32, 339, 510, 3, iters=100)) for cin in (64, 192, 768, 1536)))
