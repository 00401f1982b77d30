# usage: bash tools/w6_variants.sh name1 name2 ...   (libb2f_<name>.so built by tools/build_variant.py; "base" = libb2f.so)
for v in "$@"; do
  if [ $v = base ]; then L=back2future_amd/libb2f.so; else L=back2future_amd/libb2f_$v.so; fi
  echo "== $v"; B2F_LIB=$PWD/$L timeout 200 python tools/layer_prof.py --filter convW6 2>&1 | grep -E "total|200to128_256|128to128_256|96to64_256|232to128|32to32"
done
