# A/B on one device: RCAN bench with the one-kernel blocks and with two launches per block, alternating, three rounds.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r03f}
mkdir -p $O
cd $R
for i in 1 2 3; do
  timeout -k 10 200 python3 -u bench.py --model rcan --no-side-configs --no-cpu-baseline > $O/rcan_one_kernel_$i.json 2>/dev/null || exit 1
  timeout -k 10 200 python3 -u bench.py --model rcan --no-side-configs --no-cpu-baseline --no-fused-rcab > $O/rcan_two_launch_$i.json 2>/dev/null || exit 1
done
python3 - <<PY
import json, glob
for kind in ("one_kernel", "two_launch"):
    v = [json.load(open(f)) for f in sorted(glob.glob("$O/rcan_%s_*.json" % kind))]
    print(kind, [d["value"] for d in v], "frac", [d["roofline"]["frac"] for d in v], "serial us", [d["roofline"]["serial"]["avg_launch_us"] for d in v])
PY
