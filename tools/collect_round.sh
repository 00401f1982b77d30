#!/bin/bash
# Everything the round's evidence needs, at the current HEAD, in one GPU call:
#   bash tools/collect_round.sh r04   ->  gpurun_out/profiles_r04/*, gpurun_out/sq_r04/*  (copy the summaries into profiles/)
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
cd $R
mkdir -p gpurun_out/profiles_$TAG
python3 bench.py > gpurun_out/profiles_$TAG/${TAG}_bench_default.json 2> gpurun_out/profiles_$TAG/bench_default.err
bash tools/collect_profiles.sh $TAG > gpurun_out/profiles_$TAG/collect.log 2>&1
bash tools/collect_sq.sh $TAG > gpurun_out/profiles_$TAG/collect_sq.log 2>&1
python3 tools/sq_summary.py gpurun_out/sq_$TAG $TAG >> gpurun_out/profiles_$TAG/collect_sq.log 2>&1
cp profiles/${TAG}_sq_counters.json gpurun_out/profiles_$TAG/ 2>/dev/null
bash tools/other_shapes.sh > gpurun_out/profiles_$TAG/${TAG}_other_shapes.txt 2>&1
ls -la gpurun_out/profiles_$TAG
