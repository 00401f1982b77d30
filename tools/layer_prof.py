"""Per-layer HIP-event profile of one forward pass (GPU box only): one row per (kernel, layer shape, map size), ms per step.
    python tools/layer_prof.py [key=value ...]      # options set with b2f_set_option before the profiled passes
    python tools/layer_prof.py --batch 16 --filter convD2 s2_tiles_per_block=1
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from back2future_amd import back2future

B, H, W, flt, model = 16, 1024, 1920, "", "random:hard:2:1.0"
opts = []
a = sys.argv[1:]
while a:
    k = a.pop(0)
    if k == "--batch": B = int(a.pop(0))
    elif k == "--height": H = int(a.pop(0))
    elif k == "--width": W = int(a.pop(0))
    elif k == "--filter": flt = a.pop(0)
    elif k == "--model": model = a.pop(0)
    else:
        kk, v = k.split("=")
        opts.append((kk, int(v)))
m = back2future.Model(model)
dev = torch.device("cuda", 0)
x = bench.make_triplets(torch, B, H, W, seed=2, device=dev)
flow = torch.empty(B, 2, H, W, device=dev)
torch.cuda.synchronize()
m.set_option("profile_layers", 1)
for k, v in opts:
    m.set_option(k, v)
m.forward_device(x.data_ptr(), B, H, W, flow.data_ptr(), unit_input=True)
m.synchronize()
m.set_option("profile", 1)
m.profile_reset()
steps = 3
for _ in range(steps):
    m.forward_device(x.data_ptr(), B, H, W, flow.data_ptr(), unit_input=True)
m.synchronize()
rows = {k: (v[0] / steps, v[1] // steps) for k, v in m.profile_read().items()}
tot = sum(v[0] for v in rows.values())
print("options %s: total %.3f ms/step" % (opts, tot))
for k, (ms, n) in sorted(rows.items(), key=lambda kv: -kv[1][0]):
    if flt in k:
        print("  %-34s %8.3f ms  x%d" % (k, ms, n))
m.close()
