#!/bin/bash
# SQ counters of the conv kernels in separate rocprofv3 --pmc passes (batch 4, one step):
#   bash tools/collect_sq.sh [tag]   -> gpurun_out/sq_<tag>/passN/*counter_collection.csv  (tools/pmc_summary.py reads them)
# Never add TA_* counters (they hang rocprofv3 on this pool); every pass runs under its own timeout.
set -u
TAG=${1:-conv}
cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/sq_$TAG
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_RD"
P3="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout 240 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/pass$i -o b2f -- \
    python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --batch 16 --no-extras > $OUT.pass$i.log 2>&1
  echo "pass$i rc=$?"
done
