"""ms per frame of the RRDBNet bench for several lane counts (one line; for A/B runs)."""
import os, sys, subprocess, json
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
line = ""
extra = [a for a in sys.argv[1:] if a.startswith("--")]
for lanes in [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [1, 2, 3, 4]:
    out = subprocess.run([sys.executable, "bench.py", "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--lanes", str(lanes), *extra,
                          "--batch", str(12 if lanes == 3 else 8)], capture_output=True, text=True, cwd=root).stdout
    r = [json.loads(l) for l in out.splitlines() if l.startswith("{")][0]
    line += f" lanes{lanes} {r['ms_per_frame_per_gpu']:.3f}"
print("ms/frame:" + line)
