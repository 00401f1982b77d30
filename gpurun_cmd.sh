python -m pytest tests -m gpu -x -q 2>&1 | tail -3
mkdir -p gpurun_out/host
python tools/host_path_bench.py --n 32 > gpurun_out/host/new2_default.json 2> gpurun_out/host/new2_default.err; tail -1 gpurun_out/host/new2_default.json; tail -3 gpurun_out/host/new2_default.err
for t in 8 24 32 48; do B2F_HOST_THREADS=$t python tools/host_path_bench.py --n 32 --reps 2 > gpurun_out/host/new2_t$t.json 2>/dev/null; tail -1 gpurun_out/host/new2_t$t.json; done
for px in 4194304 16777216; do B2F_HOST_SUBBATCH_PIXELS=$px python tools/host_path_bench.py --n 32 --reps 2 > gpurun_out/host/new2_px$px.json 2>/dev/null; tail -1 gpurun_out/host/new2_px$px.json; done
python tools/host_path_bench.py --n 64 --height 375 --width 1242 --reps 2 | tail -1
python tools/host_path_bench.py --n 64 --height 436 --width 1024 --reps 2 | tail -1
