#!/bin/bash
# A/B of fused warp+cost-volume kernel variants: prints warp_costvol ms/step (batch 8).
for cfg in "1 2 0" "3 2 0"; do
  set -- $cfg
  B2F_CORR_VARIANT=$1 B2F_CORR_NK4=$2 B2F_CORR_ABLATE=$3 python bench.py --steps 2 --warmup 1 --batch 8 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('variant=$1 nk4=$2 ablate=$3', 'corr ms/step', round(d['roofline_corrwarp']['ms_per_step'], 3), 'frac', round(d['roofline_corrwarp']['frac'], 3))"
done
