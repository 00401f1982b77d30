#!/bin/bash
# Ablation of the gen-2 Winograd kernel (profiling only; results are wrong with ablate != 0).
# Variants are compile-time (a runtime branch around loads/MFMAs pessimizes s_waitcnt placement):
#   for a in 1 2 4 8 3 7 9 15; do python tools/build_variant.py wabl$a b2f_wino.hip -DB2F_WINO_ABLATE=$a; done
# built on the CPU box; the .so files travel with gpurun.
for a in 0 1 2 4 8 3 7 9 15; do
  lib=back2future_amd/libb2f_wabl$a.so
  [ $a = 0 ] && lib=back2future_amd/libb2f.so
  [ -f $lib ] || continue
  B2F_LIB=$PWD/$lib python bench.py --steps 2 --warmup 1 --batch 8 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); k=d['kernel_ms_per_step']; print('ablate=$a', 'wino_nt2', round(k['conv3x3_wino_nt2'],2), 'wino_nt1', round(k['conv3x3_wino_nt1'],2))"
done
