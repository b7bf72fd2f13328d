#!/bin/bash
# Single-frame latency of the three networks against the tile height (diagnostic build: SRMI_ROWS_PER_WAVE),
# one lane, one frame per call: tools/latency_probe.sh > gpurun_out/latency_probe.txt
export SRMI_LIB=$PWD/superresolution_amd/variants/libsrmi355_diag.so
for round in 1 2; do
for model in rrdbnet rcan edsr; do
  for mb in 3 2 4; do
    [ "$model" != rrdbnet ] && [ "$mb" = 4 ] && continue
    echo -n "$model batch 1 lanes 1 rows_per_wave $mb: "
    SRMI_ROWS_PER_WAVE=$mb timeout -k 10 120 python3 bench.py --model $model --batch 1 --lanes 1 --steps 40 --warmup 5 \
        --no-cpu-baseline --no-side-configs 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_frame_per_gpu'], 'ms', d['roofline']['frac'])" || exit 1
  done
done
done
