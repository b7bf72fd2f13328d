"""One line of timings for A/B runs: the five conv shapes of a dense block at 510x339 (us each and summed) and
the whole RRDBNet with 1 and 2 lanes (ms per frame)."""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_conv import bench
ts = []
for cin, cout in [(64, 32), (96, 32), (128, 32), (160, 32), (192, 64)]:
    us, _ = bench(cin, cout, 339, 510, 3, iters=200)
    ts.append(us)
line = "rdb convs us: " + " ".join(f"{t:5.1f}" for t in ts) + f" | sum {sum(ts):6.1f}"
for lanes in (1, 2):
    out = subprocess.run([sys.executable, "bench.py", "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--lanes", str(lanes)],
                         capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))).stdout
    r = [json.loads(l) for l in out.splitlines() if l.startswith("{")][0]
    line += f" | lanes{lanes} {r['ms_per_frame_per_gpu']:.3f} ms/frame"
print(line)
