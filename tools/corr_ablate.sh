#!/bin/bash
# Ablation of the fused warp+cost-volume kernel (profiling only): prints warp_costvol ms/step
# for B2F_CORR_ABLATE in {0, 1 (no gather loads), 2 (no FMAs), 4 (no stores), 3, 5, 6, 7}.
for a in 0 1 2 4 5 6 7; do
  B2F_CORR_ABLATE=$a python bench.py --steps 2 --warmup 1 --batch 8 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('ablate=$a', 'corr ms/step', round(d['roofline_corrwarp']['ms_per_step'], 3))"
done
