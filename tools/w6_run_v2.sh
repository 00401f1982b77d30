mkdir -p gpurun_out/w6
bash tools/w6_variants.sh base w6nbx base w6nbx > gpurun_out/w6/variants2.log 2>&1
echo "== min_pixels 4096" >> gpurun_out/w6/variants2.log
timeout 200 python tools/layer_prof.py --filter conv wino6_min_pixels=4096 2>&1 | grep -E "total|64x120" >> gpurun_out/w6/variants2.log
echo "== default" >> gpurun_out/w6/variants2.log
timeout 200 python tools/layer_prof.py --filter conv 2>&1 | grep -E "total|64x120" >> gpurun_out/w6/variants2.log
cat gpurun_out/w6/variants2.log
