# bench.py over (frames per sweep, streams per GPU); prints ms per frame
for sw in 1 2 4; do for st in 1 2 3; do
  b=$((sw*st)); [ $b -lt 8 ] && b=8; [ $st -eq 3 ] && b=$((sw*3*2))
  timeout -k 10 120 python -u bench.py --steps 4 --warmup 2 --no-cpu-baseline --sweep $sw --streams $st --batch $b 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('sweep $sw streams $st batch $b:', r['ms_per_frame_per_gpu'], 'ms/frame', r['value'], 'Mpix/s', r['roofline']['achieved'], 'TF')
" || exit 1
done; done
