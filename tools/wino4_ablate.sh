#!/bin/bash
# Ablation of the F(4x4) Winograd kernel (profiling only; results are wrong with ablate != 0).
# Compile-time variants, built on the CPU box:
#   for a in 1 2 4 8 16 7 23 31; do python tools/build_variant.py w4abl$a b2f_wino4.hip -DB2F_WINO4_ABLATE=$a; done
for a in 0 1 2 4 16 8 7 23 31; do
  lib=back2future_amd/libb2f_w4abl$a.so
  [ $a = 0 ] && lib=back2future_amd/libb2f.so
  [ -f $lib ] || continue
  B2F_LIB=$PWD/$lib python bench.py --steps 2 --warmup 1 --batch 8 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); k=d['kernel_ms_per_step']; print('ablate=$a', 'wino4', round(k['conv3x3_wino4_nt2'],2))"
done
