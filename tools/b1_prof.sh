#!/bin/bash
# Per-kernel time of a single-triplet call (the reference's own calling pattern, back2future.lua:73-74), GPU box:
#   bash tools/b1_prof.sh [H W]   ->  gpurun_out/b1prof/*_kernel_stats.csv
H=${1:-1024}; W=${2:-1920}
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/b1prof
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/b1prof -o b1 --output-format csv -- python3 $R/bench.py --batch 1 --height $H --width $W --steps 20 --warmup 3 --no-extras --no-cpu-baseline --no-host-path > $R/gpurun_out/b1prof_bench.json 2> $R/gpurun_out/b1prof.err
find $R/gpurun_out/b1prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/gpurun_out/b1_kernel_stats_${H}x${W}.csv
find $R/gpurun_out/b1prof -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $R/gpurun_out/b1_kernel_trace_${H}x${W}.csv
rm -rf $R/gpurun_out/b1prof
cut -c1-100 $R/gpurun_out/b1_kernel_stats_${H}x${W}.csv | head -5
