#!/bin/bash
# HBM traffic of the two roofline kernel classes from PMC counters, as MI355X_MICROARCH.md
# prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (TCC slots), one step
# of the default bench workload each.  Output: gpurun_out/traffic/{fetch,write}/*counter_collection.csv
# -> tools/traffic_summary.py turns them into profiles/<round>_traffic.json.
set -u
cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/traffic
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -o b2f -- \
    python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT.$c.log 2>&1
  echo "$c rc=$?"
done
