"""A/B timing of the warp + cost-volume kernel variants inside the real forward pass (GPU box only): per-level
HIP-event times of warp_costvol for every (variant, ablate) pair given on the command line, batch 16 x 3x1024x1920.
    python tools/corr_ab.py 0:0 1:0 2:0 0:1 0:2 0:4      # variant:ablate_bits (variant 0 regular, 1 latency, 2 two-pixel, 3 one direction per block, 4 window-staged; ablation: variant 0 only)
Ablated runs compute wrong results (profiling only)."""
import sys
sys.path.insert(0, ".")
import torch
import bench
from back2future_amd import back2future

import os
B, H, W = int(os.environ.get("CORR_AB_BATCH", "16")), 1024, 1920
m = back2future.Model("random:hard:2:1.0")
dev = torch.device("cuda", 0)
x = bench.make_triplets(torch, B, H, W, seed=2, device=dev)
flow = torch.empty(B, 2, H, W, device=dev)
torch.cuda.synchronize()
m.set_option("profile_layers", 1)
for spec in sys.argv[1:]:
    var, abl = (int(v) for v in spec.split(":"))
    m.set_option("corr_variant", var)
    m.set_option("corr_ablate", abl)
    m.set_option("profile", 0)
    m.forward_device(x.data_ptr(), B, H, W, flow.data_ptr(), unit_input=True)
    m.synchronize()
    m.set_option("profile", 1)
    m.profile_reset()
    steps = 3
    for _ in range(steps):
        m.forward_device(x.data_ptr(), B, H, W, flow.data_ptr(), unit_input=True)
    m.synchronize()
    rows = {k: v[0] / steps for k, v in m.profile_read().items() if k.startswith("warp_costvol")}
    tot = sum(rows.values())
    print("variant %d ablate %2d: %.3f ms/step  " % (var, abl, tot) + "  ".join("%s %.3f" % (k[13:], v) for k, v in sorted(rows.items())), flush=True)
m.close()
