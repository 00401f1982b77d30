for a in 0 59 2 32 25; do
  echo "== light trace ablate=$a"
  B2F_W1B_TRACE=16 B2F_LIB=$PWD/back2future_amd/libb2f_w1bt2a$a.so python tools/layer_prof.py --filter convV1_128to128_256 wino1d=1 2>&1 | grep -E "shader clock|convV1|loop top"
done
