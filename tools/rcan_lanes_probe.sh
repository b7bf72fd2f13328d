#!/bin/bash
# RCAN 510x339 throughput against (frames per launch, lanes): tools/rcan_lanes_probe.sh > gpurun_out/rcan_lanes.txt
for round in 1 2; do
for sweep in 1 2; do
  for lanes in 1 2 3 4; do
    echo -n "rcan sweep $sweep lanes $lanes: "
    timeout -k 10 120 python3 bench.py --model rcan --sweep $sweep --lanes $lanes --steps 8 --warmup 2 \
        --no-cpu-baseline --no-side-configs 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'Mpix/s', d['ms_per_frame_per_gpu'], 'ms', d['roofline']['frac'])" || exit 1
  done
done
done
