#!/bin/bash
# Ablation of the fused warp + cost-volume kernel (profiling only; results are wrong with ablate != 0).
# Compile-time variants, built on the CPU box:
#   for a in 1 2 4 3 5 6 7 8; do python tools/build_variant.py cabl$a b2f_corr.hip -DB2F_CORR_ABLATE=$a; done
for a in 0 1 2 4 3 5 6 7 8; do
  lib=back2future_amd/libb2f_cabl$a.so
  [ $a = 0 ] && lib=back2future_amd/libb2f.so
  [ -f $lib ] || continue
  B2F_LIB=$PWD/$lib python bench.py --steps 3 --warmup 1 --batch 8 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); k=d['kernel_ms_per_step']; print('ablate=$a', 'warp_costvol', round(k['warp_costvol'],3))"
done
