#!/bin/bash
cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/sq_lds
timeout 240 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL --output-format csv -d $OUT -o b2f -- \
  python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --batch 4 --no-cpu-baseline > $OUT.log 2>&1
echo rc=$?
