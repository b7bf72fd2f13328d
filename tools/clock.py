"""Diagnostic: the shader clock the chip holds while the bench workload runs.
Needs a -DSRMI_CLOCK build (tools/build_variant.sh clock -DSRMI_CLOCK; SRMI_LIB=superresolution_amd/variants/libsrmi355_clock.so).
Every conv workgroup of the selected layer shape records s_memtime (shader cycles) and s_memrealtime (100 MHz) at
its start and end; held clock = d(cycles) / d(real ticks) * 100 MHz, median over workgroups of the last launch,
after >= 2 s of back-to-back forwards on random frames.
usage: clock.py [model] [cin cout]     (default rrdbnet, all conv layers -> the last one that ran: conv_hr)
"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
arch = sys.argv[1] if len(sys.argv) > 1 else "rrdbnet"
if len(sys.argv) > 3:
    os.environ["SRMI_STAMP_CIN"], os.environ["SRMI_STAMP_COUT"] = sys.argv[2], sys.argv[3]
buf = torch.zeros(16 << 20, dtype=torch.int64, device="cuda")   # 128 MiB: room for the largest grid
os.environ["SRMI_STAMP_PTR"] = str(buf.data_ptr())
from bench import build_model
from superresolution_amd import _lib, synth
model, _ = build_model(arch, "f16", False, 0)
eng = model.engine()
eng.set_lanes(int(os.environ.get("LANES", "3")))
h, w, batch = (256, 256, 12) if arch == "edsr" else (339, 510, 12)
imgs = torch.from_numpy(synth.random_image(h, w, 100, n=batch)).cuda()
out = torch.empty((batch, h * eng.scale, w * eng.scale, 3), dtype=torch.uint8, device="cuda")
t0 = time.perf_counter()
steps = 0
while time.perf_counter() - t0 < float(os.environ.get("SECONDS", "3")):
    eng.forward_u8(imgs, _lib.ROUND_HALF_UP, out=out)
    torch.cuda.synchronize()
    steps += 1
wall = (time.perf_counter() - t0) / (steps * batch)
raw = buf.cpu().numpy().reshape(-1, 4).astype(np.float64)
raw = raw[(raw[:, 3] > raw[:, 1]) & (raw[:, 1] > 0)]
dc, dr = raw[:, 2] - raw[:, 0], raw[:, 3] - raw[:, 1]
mhz = dc / dr * 100.0
rec = {"model": arch, "layer_filter": sys.argv[2:4], "lanes": int(os.environ.get("LANES", "3")),
       "ms_per_frame_with_stamps": round(wall * 1e3, 3), "waves_sampled": int(len(mhz)),
       "clock_mhz_p10": round(float(np.percentile(mhz, 10)), 1), "clock_mhz_p50": round(float(np.median(mhz)), 1),
       "clock_mhz_p90": round(float(np.percentile(mhz, 90)), 1),
       "wave_lifetime_us_p50": round(float(np.median(dr)) / 100.0, 2)}
print(json.dumps(rec))
