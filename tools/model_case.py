"""ms per frame of bench.py for the given models (one line; for A/B runs): model_case.py edsr rcan rrdbnet"""
import os, sys, subprocess, json
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
line = ""
for model in sys.argv[1:]:
    out = subprocess.run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--model", model],
                         capture_output=True, text=True, cwd=root).stdout
    r = [json.loads(l) for l in out.splitlines() if l.startswith("{")][0]
    line += f" {model} {r['ms_per_frame_per_gpu']:.3f} ms ({r['roofline']['achieved']:.0f} TF)"
print(line)
