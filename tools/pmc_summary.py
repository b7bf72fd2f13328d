"""Average rocprofv3 --pmc counter values per kernel from the rocpd databases under a directory:
python tools/pmc_summary.py <dir> [kernel name filter]."""
import collections
import glob
import sqlite3
import sys

d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else "conv3x3"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(d + "/**/*.db", recursive=True):
    c = sqlite3.connect(f)
    for name, cname, val, dt in c.execute(
            "select kernel_name, counter_name, value, duration from counters_collection order by dispatch_id"):
        if flt not in name:
            continue
        k = name[:60]
        acc[k][cname].append(float(val))
        dur[k].append(float(dt))
for k, cs in acc.items():
    print(k, "launches", max(len(v) for v in cs.values()))
    for cn, v in sorted(cs.items()):
        v = v[len(v) // 2:]  # drop the warm-up half
        print(f"   {cn:36s} {sum(v) / len(v):16.1f}")
    v = dur[k][len(dur[k]) // 2:]
    print(f"   {'duration_ns':36s} {sum(v) / len(v):16.1f}")
