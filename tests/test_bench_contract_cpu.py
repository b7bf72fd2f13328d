"""The one JSON line bench.py prints (last recorded run: profiles/r02_bench_final_tree.json) carries every field the
driver's contract names, with consistent arithmetic; and bench.py refuses to run without a GPU instead of
falling back to the CPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line():
    with open(os.path.join(ROOT, "profiles", "r02_bench_final_tree.json")) as f:
        lines = [ln for ln in f.read().splitlines() if ln.strip()]
    assert len(lines) == 1, "bench.py must print exactly one line on stdout"
    return json.loads(lines[0])


def test_recorded_bench_line_has_the_contract_fields():
    d = _line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "Mpix/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f16" and "synthetic" in d["data"] and "workload" in d["config"] and "model" not in d["config"]
    assert "510x339" in d["config"]["workload"] and "510x339" in d["metric"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0 < r["frac"] < 1
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["unit"] == d["unit"] and c["cores"] >= 1
    # value = frames * output pixels / wall time of the timed steps
    frames = d["config"]["frames_per_gpu_per_step"] * d["steps"] * d["n_gpus"]
    mpix = frames * 2040 * 1356 / 1e6
    assert abs(d["value"] - mpix / (d["ms_per_step"] * d["steps"] / 1e3)) / d["value"] < 1e-3
    # algorithmic work: 6.199 TFLOP per 510x339 frame over the device time per frame
    assert abs(r["achieved"] - 6.199e12 / (d["ms_per_frame_per_gpu"] * 1e-3) / 1e12) / r["achieved"] < 2e-3


def test_bench_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        return
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], cwd=ROOT,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "GPU" in (r.stderr + r.stdout) and not r.stdout.strip().startswith("{")
