for sw in 1 2 4; do for ln in 1 2 3; do
  b=$((sw*ln*4))
  timeout -k 10 120 python -u bench.py --model edsr --steps 3 --warmup 1 --no-cpu-baseline --sweep $sw --lanes $ln --batch $b 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('edsr sweep $sw lanes $ln batch $b:', r['ms_per_frame_per_gpu'], 'ms/frame', r['roofline']['achieved'], 'TF')
" || exit 1
done; done
