# A/B of library variants on one device: bash tools/ab_lib.sh "<bench flags>" <lib a> <lib b> ...   (two alternating rounds)
R=$GRAFT_REPO_ROOT
cd $R
flags=$1; shift
for round in 1 2; do
  for lib in "$@"; do
    SRMI_LIB=$lib timeout -k 10 200 python3 -u bench.py $flags --no-side-configs --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', '$flags', d['value'], d['roofline']['frac'], 'serial', d['roofline']['serial']['frac'])" || exit 1
  done
done
