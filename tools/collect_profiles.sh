#!/bin/bash
# Round profiles: rocprofv3 kernel-trace stats of the default bench command, the bench line printed
# under the profiler, and the two PMC traffic passes.  Run on the GPU box:
#   bash tools/collect_profiles.sh r01    ->  gpurun_out/profiles_r01/...  (copy the summaries into profiles/)
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o b2f -- \
  python3 $R/bench.py --steps 5 --warmup 2 --no-extras > $OUT/bench_under_rocprof.log 2>&1
echo "stats rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -o b2f -- \
    python3 $R/bench.py --steps 1 --warmup 1 --no-extras > $OUT/pmc_$c.log 2>&1
  echo "$c rc=$?"
done
cd $R
grep '^{"metric"' $OUT/bench_under_rocprof.log | tail -1 > $OUT/${TAG}_bench_under_rocprof.json
cp $(ls $OUT/stats/*kernel_stats.csv $OUT/stats/*/*kernel_stats.csv 2>/dev/null | head -1) $OUT/${TAG}_kernel_stats.csv
mkdir -p $OUT/t/FETCH_SIZE $OUT/t/WRITE_SIZE
cp $(ls $OUT/FETCH_SIZE/*counter_collection.csv $OUT/FETCH_SIZE/*/*counter_collection.csv 2>/dev/null | head -1) $OUT/t/FETCH_SIZE/b2f_counter_collection.csv
cp $(ls $OUT/WRITE_SIZE/*counter_collection.csv $OUT/WRITE_SIZE/*/*counter_collection.csv 2>/dev/null | head -1) $OUT/t/WRITE_SIZE/b2f_counter_collection.csv
python3 tools/traffic_summary.py $OUT/t $TAG > /dev/null && cp profiles/${TAG}_traffic.json $OUT/
rm -rf $OUT/FETCH_SIZE $OUT/WRITE_SIZE $OUT/t $OUT/stats
ls -la $OUT
