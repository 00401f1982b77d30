#!/bin/bash
# PMC counters of the conv3x3_w1b launches of one forward pass (tools/layer_prof.py, batch 16, full HD), one rocprofv3 pass per counter group.
#   bash tools/w1b_pmc.sh <tag> [layer_prof options]
set -u
TAG=$1; shift
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
P1="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
P2="TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum"
P3="TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum"
P4="TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum"
P5="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/p$i -o b2f -- python3 $R/tools/layer_prof.py "$@" > $OUT/p$i.log 2>&1
done
python3 - $OUT <<'PY'
import csv, glob, sys, os
d = sys.argv[1]
for key in ("conv3x3_w1b", "conv3x3_wino4p"):
    agg, n, ms = {}, 0, 0.0
    for i in range(1, 6):
        f = glob.glob(os.path.join(d, "p%d" % i, "*counter_collection.csv")) + glob.glob(os.path.join(d, "p%d" % i, "*", "*counter_collection.csv"))
        if not f: continue
        seen, dur = set(), 0
        for r in csv.DictReader(open(f[0])):
            if key not in r["Kernel_Name"]: continue
            agg[r["Counter_Name"]] = agg.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            if r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"]); dur += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        n = max(n, len(seen)); ms = dur * 1e-6
    if not n: continue
    print("kernel *%s*: %d launches, %.3f ms in total (last pass)" % (key, n, ms))
    for k in sorted(agg): print("   %-40s %.4g" % (k, agg[k]))
    g = lambda k: agg.get(k, 0.0)
    if g("TCC_REQ_sum"): print("   L2 hit rate %.3f; EA read requests / L2 requests %.3f" % (g("TCC_HIT_sum") / max(1, g("TCC_HIT_sum") + g("TCC_MISS_sum")), g("TCC_EA0_RDREQ_sum") / g("TCC_REQ_sum")))
    if g("TCP_TCC_READ_REQ_sum"): print("   mean TCP->TCC read latency %.0f cycles; L1 accesses per L2 read request %.2f" % (g("TCP_TCC_READ_REQ_LATENCY_sum") / g("TCP_TCC_READ_REQ_sum"), g("TCP_TOTAL_CACHE_ACCESSES_sum") / g("TCP_TCC_READ_REQ_sum")))
    if g("TCP_UTCL1_TRANSLATION_HIT_sum"): print("   UTCL1 miss rate %.4f" % (g("TCP_UTCL1_TRANSLATION_MISS_sum") / (g("TCP_UTCL1_TRANSLATION_MISS_sum") + g("TCP_UTCL1_TRANSLATION_HIT_sum"))))
    if g("SQ_WAVE_CYCLES"): print("   waves: parked %.1f %%, issue-stalled %.1f %%, issuing %.1f %%; mean VMEM in flight per wave-cycle %.2f; matrix pipe busy %.3f of SIMD cycles" % (100 * g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), 100 * g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"), 100 * g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES"), g("SQ_INST_LEVEL_VMEM") / g("SQ_WAVE_CYCLES"), g("SQ_VALU_MFMA_BUSY_CYCLES") / max(1.0, g("GRBM_GUI_ACTIVE") / 8.0 * 1024)))
PY
