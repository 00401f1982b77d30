#!/usr/bin/env python3
"""Turns the three SQ passes of tools/collect_sq.sh into profiles/<tag>_sq_counters.json: per kernel (template instantiation)
and per bench step -- matrix-pipe busy fraction, VALU-class instructions per MFMA, where the wave cycles go (parked in
s_waitcnt / s_barrier, stalled at issue, issuing), LDS bank-conflict share.

    python tools/sq_summary.py gpurun_out/sq_<tag> <tag>

Units (MI355X_MICROARCH.md, rocprofv3 PMC section): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over
waves, SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs.  Every forward pass
of `bench.py --no-extras` is the same pass, so totals / (number of conv_first launches) = per step."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = name.replace("void ", "").replace("b2f::", "")
    return re.sub(r"\(.*", "", name)


def read_pass(d):
    files = glob.glob(os.path.join(d, "*counter_collection.csv")) + glob.glob(os.path.join(d, "*", "*counter_collection.csv"))
    agg, dur, cnt, seen = defaultdict(lambda: defaultdict(float)), defaultdict(float), defaultdict(int), set()
    with open(files[0]) as f:
        for r in csv.DictReader(f):
            if "b2f::" not in r["Kernel_Name"]:
                continue
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp = int(r["Dispatch_Id"])
            if disp not in seen:
                seen.add(disp)
                dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                cnt[k] += 1
    return agg, dur, cnt


def main():
    base, tag = sys.argv[1], sys.argv[2]
    passes = [read_pass(os.path.join(base, "pass%d" % i)) for i in (1, 2, 3)]
    steps = [max(1, sum(c for k, c in p[2].items() if "conv_first" in k)) for p in passes]
    out = {"source": "rocprofv3 --kernel-trace --pmc, three separate passes of `bench.py --steps 1 --warmup 1 --batch 16 --no-extras` "
                     "(tools/collect_sq.sh); per bench step", "kernels": {}}
    for k in sorted(passes[0][0], key=lambda k: -passes[0][1][k]):
        c = {}
        for (agg, dur, cnt), st in zip(passes, steps):
            for name, v in agg.get(k, {}).items():
                c[name] = v / st
        st0 = steps[0]
        e = {"launches_per_step": passes[0][2][k] / st0, "ms_per_step": passes[0][1][k] / st0 / 1e6}
        simd_cycles = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0 * 256 * 4            # per XCD active cycles x all SIMDs of the chip
        if c.get("SQ_INSTS_MFMA"):
            e["mfma_insts"] = c["SQ_INSTS_MFMA"]
            e["valu_class_insts_per_mfma"] = (c.get("SQ_INSTS_VALU", 0.0) - c["SQ_INSTS_MFMA"]) / c["SQ_INSTS_MFMA"]
            if simd_cycles:
                e["mfma_busy_frac_of_simd_cycles"] = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / simd_cycles
        wc = c.get("SQ_WAVE_CYCLES")
        if wc:
            # SQ_WAVE_CYCLES comes from pass 1, the SQ_WAIT_* / ACTIVE counters from pass 2 / 3 (other runs of the same step)
            e["wave_cycles_parked_waitcnt_barrier"] = c.get("SQ_WAIT_ANY", 0.0) / wc
            e["wave_cycles_issue_stalled"] = c.get("SQ_WAIT_INST_ANY", 0.0) / wc
            e["wave_cycles_issuing"] = c.get("SQ_ACTIVE_INST_ANY", 0.0) / wc
        if c.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_bank_conflict_share_of_lds_cycles"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]
        e["vmem_insts"] = c.get("SQ_INSTS_VMEM", 0.0)
        e["lds_insts"] = c.get("SQ_INSTS_LDS", 0.0)
        e["valu_insts"] = c.get("SQ_INSTS_VALU", 0.0)
        out["kernels"][k] = e
    path = "profiles/%s_sq_counters.json" % tag
    json.dump(out, open(path, "w"), indent=1)
    print(path)


if __name__ == "__main__":
    main()
