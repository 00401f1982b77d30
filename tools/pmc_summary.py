#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel: sums each counter over
dispatches (optionally only the largest dispatch of each kernel) and prints duration too.

    python tools/pmc_summary.py gpurun_out/pmc1/b2f_counter_collection.csv [--filter b2f] [--top]
"""
import csv
import sys
from collections import defaultdict


def short(name):
    name = name.replace("void ", "").replace("b2f::", "")
    return name.split("(")[0][:48]


def main():
    path = sys.argv[1]
    filt = sys.argv[sys.argv.index("--filter") + 1] if "--filter" in sys.argv else "b2f::"
    per_dispatch = defaultdict(dict)
    meta = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            if filt not in r["Kernel_Name"]:
                continue
            d = int(r["Dispatch_Id"])
            per_dispatch[d][r["Counter_Name"]] = per_dispatch[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            meta[d] = (short(r["Kernel_Name"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r["Grid_Size"]),
                       r["VGPR_Count"], r["LDS_Block_Size"])
    agg = defaultdict(lambda: defaultdict(float))
    dur = defaultdict(float)
    cnt = defaultdict(int)
    for d, ctrs in per_dispatch.items():
        k = meta[d][0]
        if "--top" in sys.argv:
            k = "%s grid=%d" % (k, meta[d][2])
        for c, v in ctrs.items():
            agg[k][c] += v
        dur[k] += meta[d][1]
        cnt[k] += 1
    names = sorted({c for k in agg for c in agg[k]})
    print("kernel".ljust(58), "n".rjust(4), "dur_us".rjust(10), " ".join(n.replace("SQ_", "").rjust(16) for n in names))
    for k in sorted(agg, key=lambda k: -dur[k]):
        print(k.ljust(58), str(cnt[k]).rjust(4), ("%.1f" % (dur[k] / 1e3)).rjust(10),
              " ".join(("%.4g" % agg[k][n]).rjust(16) for n in names))


if __name__ == "__main__":
    main()
