#!/bin/bash
# SQ / LDS counters of the warp + cost-volume kernel variants (separate rocprofv3 --pmc passes, no TA_* counters):
#   bash tools/collect_corr_pmc.sh "0:0 3:0"   -> gpurun_out/corrpmc/passN/... ; summary printed per kernel and grid size
set -u
SPECS=${1:-"0:0 3:0"}
cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/corrpmc
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD"
P3="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  ( cd $GRAFT_REPO_ROOT && timeout 200 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/pass$i -o b2f -- python3 tools/corr_ab.py $SPECS > $OUT.pass$i.log 2>&1 )
  echo "pass$i rc=$?"
  python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $(ls $OUT/pass$i/*counter_collection.csv $OUT/pass$i/*/*counter_collection.csv 2>/dev/null | head -1) --filter warp_costvol --top | cut -c1-260
done
