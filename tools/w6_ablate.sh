mkdir -p gpurun_out/w6
for a in 0 128 32 160 2; do
  if [ $a = 0 ]; then L=back2future_amd/libb2f.so; else L=back2future_amd/libb2f_w6a$a.so; fi
  echo "== ablate $a"; B2F_LIB=$PWD/$L timeout 200 python tools/layer_prof.py --filter convW6 wino6=1 2>&1 | grep -E "total|200to128_256|128to128_256|96to64_256|232to128"
done
