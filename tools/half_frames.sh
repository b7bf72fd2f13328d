# lane scaling on half-height frames (diagnostic; needs an engine built with kMaxLanes >= 8)
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-8}
for hgt in 170; do for ln in 3 4 5 6 7 8; do
  timeout -k 10 120 python -u bench.py --steps 3 --warmup 1 --no-cpu-baseline --height $hgt --lanes $ln --batch $((ln*6)) 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('queues $GPU_MAX_HW_QUEUES height $hgt lanes $ln:', r['ms_per_frame_per_gpu'], 'ms/frame', r['roofline']['achieved'], 'TF')
"; done; done
