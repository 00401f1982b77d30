#!/bin/bash
# HBM-side traffic of the warp + cost-volume kernel variants per launch (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes):
#   bash tools/collect_corr_fetch.sh "3:0 6:0"
# FETCH_SIZE is in 64-B units of which gfx950 tallies 128-B requests as one: x 2 (MI355X_MICROARCH.md); printed raw here.
set -u
SPECS=${1:-"3:0 6:0"}
cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/corrfetch
i=0
for P in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  ( cd $GRAFT_REPO_ROOT && timeout 200 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/pass$i -o b2f -- python3 tools/corr_ab.py $SPECS > $OUT.pass$i.log 2>&1 )
  echo "pass$i rc=$?"
  python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $(ls $OUT/pass$i/*counter_collection.csv $OUT/pass$i/*/*counter_collection.csv 2>/dev/null | head -1) --filter warp_costvol --top | cut -c1-200
done
