"""One line of timings for A/B runs (diagnostic builds honour SRMI_ROWS_PER_WAVE): the five conv shapes of a dense block at
510x339 (us each and summed, rows per wave from $MB, default 3) and the whole RRDBNet bench with 1 and 3 lanes."""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_conv import bench
mb = int(os.environ.get("MB", "3"))
ts = []
for cin, cout in [(64, 32), (96, 32), (128, 32), (160, 32), (192, 64)]:
    us, _ = bench(cin, cout, 339, 510, mb, iters=200)
    ts.append(us)
line = f"mb{mb} rdb convs us: " + " ".join(f"{t:5.1f}" for t in ts) + f" | sum {sum(ts):6.1f}"
for lanes in (1, 3):
    out = subprocess.run([sys.executable, "bench.py", "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--no-side-configs",
                          "--lanes", str(lanes)] + sys.argv[1:],
                         capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))).stdout
    r = [json.loads(l) for l in out.splitlines() if l.startswith("{")][0]
    line += f" | lanes{lanes} {r['ms_per_frame_per_gpu']:.3f} ms/frame"
print(line, flush=True)
