# A/B of one bench.py flag on one device, two alternating rounds: bash tools/ab_flag.sh "<common flags>" "<flag under test>"
cd $GRAFT_REPO_ROOT
for round in 1 2; do
  for f in "" "$2"; do
    timeout -k 10 200 python3 -u bench.py $1 $f --no-side-configs --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('[$1] [$f]', d['value'], d['roofline']['frac'], 'serial', d['roofline']['serial']['frac'])" || exit 1
  done
done
