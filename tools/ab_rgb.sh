# A/B of the RGB layer's kernel inside the whole networks on one device (two alternating rounds):
# bash tools/ab_rgb.sh   (SRMI_RGB_GATHER=0 keeps the layer on the direct kernel)
cd $GRAFT_REPO_ROOT
for model in rrdbnet edsr rcan; do
  for round in 1 2; do
    for g in 0 1; do
      SRMI_RGB_GATHER=$g timeout -k 10 200 python3 -u bench.py --model $model --no-side-configs --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$model', 'gather=$g', d['value'], d['roofline']['frac'])" || exit 1
    done
  done
done
