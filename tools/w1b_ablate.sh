#!/bin/bash
# Ablation of the 1-D Winograd bf16 kernel (profiling only; results are wrong with ablate != 0).  Variants built on the CPU box:
#   for a in 2 32 34 25 4 64 1 8 16; do python tools/build_variant.py w1babl$a b2f_w1b.hip -DB2F_W1B_ABLATE=$a; done
# bits: 1 no raw loads, 2 no weight loads, 4 no MFMAs, 8 no transform / split, 16 no V writes, 32 no pixel-window reads, 64 no epilogue stores
for a in 0 2 32 34 25 59 4 64 1 8 16; do
  lib=back2future_amd/libb2f_w1babl$a.so
  [ $a = 0 ] && lib=back2future_amd/libb2f.so
  [ -f $lib ] || continue
  echo "ablate=$a: $(B2F_LIB=$PWD/$lib python tools/layer_prof.py --filter convV1 wino1d=1 2>/dev/null | grep -E '200to128_256|128to128_256|96to64_256|32to32_256' | tr -s ' ' | tr '\n' ';')"
done
