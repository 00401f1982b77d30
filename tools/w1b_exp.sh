for o in "w1b_store_aux=0" "w1b_store_aux=1" "w1b_store_aux=2" "w1b_store_aux=3"; do
  echo "$o: $(python tools/layer_prof.py --filter convV1 wino1d=1 $o 2>/dev/null | grep -E 'total|200to128_256|128to128_256|96to64_256|32to32_256|128to128_128' | tr -s ' ' | tr '\n' ';')"
done
