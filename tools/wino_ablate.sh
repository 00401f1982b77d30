#!/bin/bash
# Ablation of the Winograd conv kernel (profiling only; results are wrong with ablate != 0).
for a in 0 16 1 2 4 8 3 7 9 15; do
  B2F_WINO_ABLATE=$a python bench.py --steps 2 --warmup 1 --batch 8 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); k=d['kernel_ms_per_step']; print('ablate=$a', 'wino_nt2', round(k['conv3x3_wino_nt2'],2), 'wino_nt1', round(k['conv3x3_wino_nt1'],2))"
done
