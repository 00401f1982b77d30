#!/bin/bash
# L2 / fabric-side counters (TCC_*, TCP_TCC_*) of every kernel of a bench step, in separate rocprofv3 --pmc passes (4 TCC slots per
# pass; never TA_* counters: they hang rocprofv3 on this pool):
#   bash tools/collect_tcc.sh [tag]   -> gpurun_out/tcc_<tag>/passN/...; python3 tools/pmc_summary.py <csv> prints them per kernel
set -u
TAG=${1:-r03}
cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/tcc_$TAG
P1="TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
P2="TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum"
P3="TCC_BUSY_sum TCC_CYCLE_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_RDREQ_32B_sum"
P4="TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum"
P5="TCC_EA0_WRREQ_64B_sum TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/pass$i -o b2f -- \
    python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --batch 16 --no-extras > $OUT.pass$i.log 2>&1
  echo "pass$i rc=$?"
  python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $(ls $OUT/pass$i/*counter_collection.csv $OUT/pass$i/*/*counter_collection.csv 2>/dev/null | head -1) | cut -c1-200 | head -14
done
