#!/bin/bash
# Ablation of the fused warp + cost-volume kernel (profiling only; results are wrong with ablate != 0).
# Runtime flags (env B2F_CORR_ABLATE, read by the launcher): 1 no gather loads, 2 no FMAs, 4 no stores, 8 no XCD remap.
# (A compile-time variant of these switches was tried and reverted: it changed register allocation -- 9 spilled
#  VGPRs at the 168-register occupancy limit -- and made the kernel 14 % slower.)
for a in 0 1 2 4 3 5 6 7 8; do
  B2F_CORR_ABLATE=$a python bench.py --steps 3 --warmup 1 --batch 8 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); k=d['kernel_ms_per_step']; print('ablate=$a', 'warp_costvol', round(k['warp_costvol'],3))"
done
