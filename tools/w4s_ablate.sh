#!/bin/bash
# Ablation of the split-operand F(4x4) kernel (profiling only; results are wrong with ablate != 0).  Variants built on the CPU box:
#   for a in 1 2 4 8 16 6; do python tools/build_variant.py w4sabl$a b2f_wino4s.hip -DB2F_W4S_ABLATE=$a; done
# bits: 1 no input transform, 2 no raw staging, 4 no B loads, 8 no MFMAs, 16 no split
for a in 0 1 2 4 6 8 16; do
  lib=back2future_amd/libb2f_w4sabl$a.so
  [ $a = 0 ] && lib=back2future_amd/libb2f.so
  [ -f $lib ] || continue
  echo "ablate=$a: $(B2F_LIB=$PWD/$lib python tools/layer_prof.py --batch 8 --filter convW4 wino4_split=1 2>/dev/null | grep -E '200to128_256|128to128_256|96to64_256' | tr -s ' ' | tr '\n' ';')"
done
