#!/bin/bash
# Throughput against frame size near the point where the lanes' working set crosses the 256 MiB Infinity Cache:
#   bash tools/size_probe.sh <model> "<H>x<W>x<batch>[xlanes] ..."   -> one line per case
model=$1; shift
for c in $1; do
  IFS=x read h w b l <<< "$c"
  l=${l:-3}
  line=$(timeout -k 10 120 python3 bench.py --model $model --steps 8 --warmup 2 --no-cpu-baseline --no-side-configs --height $h --width $w --batch $b --lanes $l 2>/dev/null) || exit 1
  echo "$model ${h}x${w} batch $b lanes $l: $(echo "$line" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], "Mpix/s", d["ms_per_frame_per_gpu"], "ms/frame frac", d["roofline"]["frac"], "one-lane", d["roofline"].get("serial",{}).get("frac"))')"
done
