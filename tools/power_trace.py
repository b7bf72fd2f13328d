"""Socket power and shader clock while a workload runs.

    python tools/power_trace.py OUT.jsonl -- python bench.py --steps 60 --no-cpu-baseline --no-side-configs

Starts the command as a child process and, every 50 ms until it exits, reads what the amdgpu driver
publishes for the first GPU under /sys/class/drm/card*/device: hwmon power (power1_average or
power1_input, microwatts), the power cap (power1_cap), the shader clock (freq1_input, Hz), the
memory clock (freq2_input), the junction temperature, gpu_busy_percent and mem_busy_percent; the shader clock also
from `rocm-smi --showclocks` every few hundred milliseconds (hwmon has no clock file on every driver).  One JSON object per
sample goes to OUT.jsonl, and a summary (idle level, level while the GPU was busy, cap) is printed
at the end.  Reading these files needs no privileges and touches no GPU state.
"""
import glob
import json
import os
import re
import subprocess
import sys
import threading
import time


def _first(patterns):
    for p in patterns:
        for f in sorted(glob.glob(p)):
            try:
                with open(f) as fh:
                    fh.read()
                return f
            except OSError:
                continue
    return None


def _read(path):
    if path is None:
        return None
    try:
        with open(path) as fh:
            return int(fh.read().split()[0])
    except (OSError, ValueError, IndexError):
        return None


def sources(base):
    hw = base + "/hwmon/hwmon*"
    return {
        "power_uW": _first([hw + "/power1_average", hw + "/power1_input"]),
        "cap_uW": _first([hw + "/power1_cap"]),
        "cap_max_uW": _first([hw + "/power1_cap_max"]),
        "sclk_Hz": _first([hw + "/freq1_input"]),
        "mclk_Hz": _first([hw + "/freq2_input"]),
        "temp_mC": _first([hw + "/temp2_input", hw + "/temp1_input"]),
        "busy_pct": _first([base + "/gpu_busy_percent"]),
        "mem_busy_pct": _first([base + "/mem_busy_percent"]),
    }


def main():
    if "--" not in sys.argv or len(sys.argv) < 4:
        sys.exit(__doc__)
    out = sys.argv[1]
    cmd = sys.argv[sys.argv.index("--") + 1:]
    # every card of the host is sampled (only one is visible to the child); the summary is of the card
    # whose power rose most while the child ran
    cards = {os.path.basename(os.path.dirname(b)): sources(b) for b in sorted(glob.glob("/sys/class/drm/card*/device"))
             if re.fullmatch(r"card\d+", os.path.basename(os.path.dirname(b)))}
    cards = {c: s for c, s in cards.items() if s["power_uW"] is not None}
    if not cards:
        sys.exit("no card with a readable hwmon power file")
    print("cards:", json.dumps(cards), file=sys.stderr)
    samples = []
    t0 = time.perf_counter()

    def take(phase):
        for c, src in cards.items():
            samples.append({"t": round(time.perf_counter() - t0, 3), "phase": phase, "card": c,
                            **{k: _read(p) for k, p in src.items()}})

    for _ in range(10):                      # idle level before the child touches the GPU
        take("idle")
        time.sleep(0.05)
    child = subprocess.Popen(cmd)
    sclk = []                                # (t, MHz) from rocm-smi: hwmon publishes no clock on this driver
    stop = threading.Event()

    def clocks():
        while not stop.is_set():
            try:
                r = subprocess.run(["rocm-smi", "--showclocks", "--json"], capture_output=True, text=True, timeout=10)
                for dev in json.loads(r.stdout[r.stdout.index("{"):]).values():
                    m = re.search(r"(\d+)Mhz", dev.get("sclk clock speed:", ""))
                    if m:
                        sclk.append((round(time.perf_counter() - t0, 3), int(m.group(1))))
            except Exception:
                pass
            stop.wait(0.2)

    th = threading.Thread(target=clocks, daemon=True)
    th.start()
    while child.poll() is None:
        take("run")
        time.sleep(0.05)
    stop.set()
    th.join(timeout=15)

    def rise(c):
        r = [s["power_uW"] for s in samples if s["card"] == c and s["phase"] == "run" and s["power_uW"] is not None]
        i = [s["power_uW"] for s in samples if s["card"] == c and s["phase"] == "idle" and s["power_uW"] is not None]
        return (max(r) if r else 0) - (min(i) if i else 0)

    card = max(cards, key=rise)
    src = cards[card]
    samples_all, samples = samples, [s for s in samples if s["card"] == card]
    with open(out, "w") as fh:
        for s in samples_all:
            fh.write(json.dumps(s) + "\n")

    def stat(rows, key, scale):
        v = sorted(r[key] * scale for r in rows if r.get(key) is not None)
        if not v:
            return None
        return {"min": round(v[0], 1), "median": round(v[len(v) // 2], 1), "p90": round(v[int(len(v) * 0.9)], 1), "max": round(v[-1], 1)}

    idle = [s for s in samples if s["phase"] == "idle"]
    run = [s for s in samples if s["phase"] == "run"]
    busy = [s for s in run if (s.get("busy_pct") or 0) >= 90] or run
    summary = {
        "card": card, "samples": len(samples), "busy_samples": len(busy),
        "cap_W": (_read(src["cap_uW"]) or 0) / 1e6 or None,
        "cap_max_W": (_read(src["cap_max_uW"]) or 0) / 1e6 or None,
        "idle_power_W": stat(idle, "power_uW", 1e-6), "busy_power_W": stat(busy, "power_uW", 1e-6),
        "idle_sclk_MHz": stat(idle, "sclk_Hz", 1e-6), "busy_sclk_MHz": stat(busy, "sclk_Hz", 1e-6),
        "busy_mclk_MHz": stat(busy, "mclk_Hz", 1e-6), "busy_temp_C": stat(busy, "temp_mC", 1e-3),
        "busy_mem_busy_pct": stat(busy, "mem_busy_pct", 1),
        "rocm_smi_sclk_MHz_while_busy": sorted(v for t, v in sclk if v >= 500),
        "command": " ".join(cmd), "child_rc": child.returncode,
    }
    print(json.dumps(summary))
    sys.exit(child.returncode)


if __name__ == "__main__":
    main()
