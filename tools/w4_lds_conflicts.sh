#!/bin/bash
# LDS bank conflicts of the persistent F(4x4) kernel by component: SQ counters of ONE layer (128 -> 128 at 128 x 240, batch 4)
# for the regular library and the ablation builds (results of those are wrong).  Variants, built on the CPU box:
#   for a in 1 2 16 64 128 256 512; do python tools/build_variant.py w4abl$a b2f_wino4.hip -DB2F_WINO4_ABLATE=$a; done
set -u
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04/ldsconf
mkdir -p $OUT
for a in 0 1 2 16 64 128 256 512; do
  lib=$R/back2future_amd/libb2f_w4abl$a.so
  [ $a = 0 ] && lib=$R/back2future_amd/libb2f.so
  [ -f $lib ] || continue
  B2F_LIB=$lib timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/a$a -o b2f -- python3 $R/tools/one_layer.py > $OUT/a$a.log 2>&1
  python3 - $OUT/a$a $a <<'PY'
import csv, glob, sys, os
d, a = sys.argv[1], sys.argv[2]
f = glob.glob(os.path.join(d, "*counter_collection.csv")) + glob.glob(os.path.join(d, "*", "*counter_collection.csv"))
agg, n, dur, seen = {}, 0, 0, set()
for r in csv.DictReader(open(f[0])):
    if "wino4p" not in r["Kernel_Name"]:
        continue
    agg[r["Counter_Name"]] = agg.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    if r["Dispatch_Id"] not in seen:
        seen.add(r["Dispatch_Id"]); dur += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
n = max(1, len(seen))
print("ablate=%-4s launches %d  %.3f ms  LDS insts %.3gM  idx_active %.4gM  bank_conflict %.4gM (%.1f %% of idx_active)  active_inst_lds %.4gM" % (
    a, n, dur / n * 1e-6, agg.get("SQ_INSTS_LDS", 0) / n / 1e6, agg.get("SQ_LDS_IDX_ACTIVE", 0) / n / 1e6, agg.get("SQ_LDS_BANK_CONFLICT", 0) / n / 1e6,
    100.0 * agg.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, agg.get("SQ_LDS_IDX_ACTIVE", 1)), agg.get("SQ_ACTIVE_INST_LDS", 0) / n / 1e6))
PY
done
