#!/bin/bash
# SQ counters of ONE conv layer (tools/head_run.py; options through B2F_ONE_LAYER_OPTS) in two rocprofv3 --pmc passes:
#   bash tools/one_layer_sq.sh <kernel name substring> [ci co h w batch]      e.g.  B2F_ONE_LAYER_OPTS=wino2_split=1 bash tools/one_layer_sq.sh wino2s
set -u
K=$1; shift
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04/sq_head_$K
mkdir -p $OUT
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS"
P3="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/p$i -o b2f -- python3 $R/tools/head_run.py "$@" > $OUT/p$i.log 2>&1
done
python3 - $OUT $K <<'PY'
import csv, glob, sys, os
d, key = sys.argv[1], sys.argv[2]
agg, n = {}, 0
for i in (1, 2, 3):
    f = glob.glob(os.path.join(d, "p%d" % i, "*counter_collection.csv")) + glob.glob(os.path.join(d, "p%d" % i, "*", "*counter_collection.csv"))
    seen, dur = set(), 0
    for r in csv.DictReader(open(f[0])):
        if key not in r["Kernel_Name"]:
            continue
        agg[r["Counter_Name"]] = agg.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); dur += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    n = max(1, len(seen))
    agg["ms_p%d" % i] = dur / n * 1e-6
g = lambda k: agg.get(k, 0.0) / n
wc = g("SQ_WAVE_CYCLES")
print("kernel *%s*: %d launches, %.3f ms" % (key, n, agg["ms_p1"]))
print("  insts per launch (M): VALU %.2f  MFMA %.3f  VMEM %.3f  LDS %.2f  SALU %.2f   VALU per MFMA %.1f" % (g("SQ_INSTS_VALU") / 1e6, g("SQ_INSTS_MFMA") / 1e6, g("SQ_INSTS_VMEM") / 1e6, g("SQ_INSTS_LDS") / 1e6, g("SQ_INSTS_SALU") / 1e6, (g("SQ_INSTS_VALU") - g("SQ_INSTS_MFMA")) / max(1.0, g("SQ_INSTS_MFMA"))))
print("  wave cycles (quad-cycles): parked at waitcnt / barrier %.1f %%, stalled at issue %.1f %% (of which LDS issue %.1f %%), issuing %.1f %%" % (100 * g("SQ_WAIT_ANY") / wc, 100 * g("SQ_WAIT_INST_ANY") / wc, 100 * g("SQ_WAIT_INST_LDS") / wc, 100 * g("SQ_ACTIVE_INST_ANY") / wc))
simd = g("GRBM_GUI_ACTIVE") / 8.0 * 256 * 4
print("  matrix pipe busy %.1f %% of SIMD cycles (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs))" % (100 * g("SQ_VALU_MFMA_BUSY_CYCLES") / max(1.0, simd)))
print("  active VALU %.1f %%, VMEM %.1f %%, LDS %.1f %% of wave cycles; LDS bank conflicts %.1f %% of LDS cycles" % (100 * g("SQ_ACTIVE_INST_VALU") / wc, 100 * g("SQ_ACTIVE_INST_VMEM") / wc, 100 * g("SQ_ACTIVE_INST_LDS") / wc, 100 * g("SQ_LDS_BANK_CONFLICT") / max(1.0, g("SQ_LDS_IDX_ACTIVE"))))
PY
