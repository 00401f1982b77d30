"""profiles/<tag>_other_shapes.json from the bench lines of tools/other_shapes.sh:
    python tools/other_shapes_summary.py gpurun_out/profiles_r04/r04_other_shapes.txt r04"""
import json
import sys

src, tag = sys.argv[1], sys.argv[2]
rows = []
for line in open(src):
    if not line.startswith('{"metric"'):
        continue
    d = json.loads(line)
    c = d["config"]
    b = c["batch_per_gpu"]
    rows.append({"batch": b, "height": c["height"], "width": c["width"], "triplets_per_s": round(d["value"], 1), "ms_per_step": round(d["ms_per_step"], 3),
                 "ms_per_triplet": round(d["ms_per_step"] / b, 3), "roofline_frac_dominant_kernel": round(d["roofline"]["frac"], 3),
                 "dominant_kernel": d["roofline"]["kernel"][:24], "roofline_corrwarp_frac": round(d["roofline_corrwarp"]["frac"], 3)})
lat = {"%dx%d" % (r["height"], r["width"]): r["ms_per_step"] for r in rows if r["batch"] == 1}
out = {"source": "tools/other_shapes.sh at the round's HEAD: bench.py --batch B --height H --width W --steps 5 --warmup 2 --no-extras (device-resident, "
                 "hipGraph replay), one MI355X", "rows": rows, "single_triplet_latency_ms": lat}
json.dump(out, open("profiles/%s_other_shapes.json" % tag, "w"), indent=1)
print(json.dumps(lat))
