# bench.py over (frames per sweep, lanes); prints ms per frame
for sw in 1 2 3; do for ln in 1 2 3 4; do
  b=$((sw*ln*2)); [ $b -lt 8 ] && b=$((sw*ln*4))
  timeout -k 10 120 python -u bench.py --steps 4 --warmup 2 --no-cpu-baseline --sweep $sw --lanes $ln --batch $b 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('sweep $sw lanes $ln batch $b:', r['ms_per_frame_per_gpu'], 'ms/frame', r['value'], 'Mpix/s', r['roofline']['achieved'], 'TF')
" || exit 1
done; done
