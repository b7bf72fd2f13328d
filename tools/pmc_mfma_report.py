"""MFMA-pipe utilisation per conv kernel from one rocprofv3 --pmc pass (rocpd databases under <dir>):
python tools/pmc_mfma_report.py <dir> <network>.

busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x SQ_BUSY_CYCLES / 32 shader engines): the share of SIMD-cycles, while
the kernel is resident, in which the matrix pipe works.  The last line is JSON: the duration-weighted figure over all
conv launches (what bench.py shows as roofline.mfma_busy) and the per-kernel figures."""
import collections
import glob
import json
import sqlite3
import sys

d, net = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(d + "/**/*.db", recursive=True):
    c = sqlite3.connect(f)
    for name, cname, val, dt in c.execute(
            "select kernel_name, counter_name, value, duration from counters_collection order by dispatch_id"):
        if "conv3x3" not in name and "rcab" not in name and "conv_rgb" not in name:
            continue
        k = name.replace("void srmi::", "").replace("(anonymous namespace)::", "").replace("(srmi::ConvParams, int)", "").replace("(srmi::ConvParams)", "")[:60]
        acc[k][cname].append(float(val))
        dur[k].append(float(dt))
print(f"# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY "
      f"GRBM_GUI_ACTIVE -- python3 bench.py --model {net} --steps 1 --warmup 1 --batch 2 --lanes 1 (tools/pmc_mfma_all.sh)")
print("# busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 x SQ_BUSY_CYCLES / 32); second half of the launches (warm-up dropped)")
tot_m = tot_b = tot_t = 0.0
per = {}
for k, cs in sorted(acc.items()):
    n = max(len(v) for v in cs.values())
    half = {cn: v[len(v) // 2:] for cn, v in cs.items()}
    t = dur[k][len(dur[k]) // 2:]
    m, b = sum(half["SQ_VALU_MFMA_BUSY_CYCLES"]), sum(half["SQ_BUSY_CYCLES"])
    busy = m / (1024.0 * b / 32.0) if b else 0.0
    per[k] = {"launches": n, "avg_us": round(sum(t) / len(t) / 1e3, 2), "mfma_busy": round(busy, 4)}
    print(f"{k:62s} launches {n:6d}  avg {sum(t) / len(t) / 1e3:8.2f} us  mfma_busy {busy:.3f}")
    for cn, v in sorted(half.items()):
        print(f"      {cn:32s} {sum(v) / len(v):16.1f}")
    tot_m += m
    tot_b += b
    tot_t += sum(t)
overall = tot_m / (1024.0 * tot_b / 32.0) if tot_b else 0.0
print(json.dumps({"network": net, "mfma_busy": round(overall, 4), "kernels": per}))
