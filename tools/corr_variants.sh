#!/bin/bash
# A/B of fused warp+cost-volume kernel variants: prints warp_costvol ms/step.
for v in 1 5; do
  B2F_CORR_VARIANT=$v python bench.py --steps 2 --warmup 1 --batch ${BATCH:-16} --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('variant=$v', 'corr ms/step', round(d['roofline_corrwarp']['ms_per_step'], 3), 'frac', round(d['roofline_corrwarp']['frac'], 3))"
done
