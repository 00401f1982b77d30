"""GPU box: the Winograd F(6x6,3x3) kernel (option wino6 = 1, b2f_wino6.hip) against an fp64 convolution and the F(4x4) kernel (wino6 = 0).
    python tools/wino6_check.py [seed] [ncases]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from back2future_amd import back2future, ops

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
nrand = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rng = np.random.default_rng(seed)
m = back2future.Model("random:hard:1:1.0")
m.set_option("wino6_min_pixels", 0)
bad = 0
# (B, ci, co, h, w, scale, blocks): shapes of the graph at small sizes, ragged edges, odd / even chunk counts, > 1 n-block, many items per block
cases = [(1, 32, 64, 12, 48, 1.0, 1), (1, 32, 64, 24, 96, 1.0, 1), (2, 128, 128, 32, 64, 1.0, 1), (1, 200, 128, 37, 71, 1.0, 1), (1, 128, 96, 64, 120, 20.0, 1),
         (3, 96, 64, 9, 130, 1e-2, 1), (1, 64, 64, 33, 50, 1.0, 1), (2, 40, 100, 20, 20, 1.0, 1), (1, 64, 64, 1, 1, 1.0, 1), (1, 32, 36, 5, 3, 1.0, 1),
         (4, 64, 64, 48, 96, 1.0, 3), (4, 40, 128, 40, 70, 1.0, 5), (2, 104, 192, 48, 33, 1.0, 7), (1, 232, 128, 50, 100, 1.0, 4),
         (2, 64, 32, 40, 100, 1.0, 1), (3, 32, 32, 50, 70, 1.0, 3), (1, 128, 96, 30, 64, 1.0, 2), (2, 72, 160, 25, 49, 1.0, 5)]
for _ in range(nrand):
    cases.append((int(rng.integers(1, 4)), int(rng.integers(4, 26)) * 8, int(rng.integers(9, 49)) * 4, int(rng.integers(1, 70)), int(rng.integers(1, 140)),
                  float(10.0 ** rng.integers(-3, 3)), int(rng.choice([1, 1, 2, 7]))))
for (B, ci, co, h, w, scale, blocks) in cases:
    x = (rng.standard_normal((B, ci, h, w)) * scale).astype(np.float32)
    wt = (rng.standard_normal((co, ci, 3, 3)) / np.sqrt(9 * ci)).astype(np.float32)
    b = (rng.standard_normal(co) * scale).astype(np.float32)
    leaky = bool(rng.integers(2))
    res = {}
    for opt in (0, 1):
        m.set_option("wino6", opt)
        m.set_option("wino4_persistent", blocks)
        res[opt] = ops.conv3x3(m, x, wt, b, 1, leaky)
    m.set_option("wino4_persistent", 1)
    full = ops.conv3x3(m, x, wt, b, 1, leaky)
    m.set_option("wino6", 0)
    y = torch.nn.functional.conv2d(torch.from_numpy(x).double(), torch.from_numpy(wt).double(), torch.from_numpy(b).double(), padding=1)
    e = (torch.where(y > 0, y, 0.2 * y) if leaky else y).numpy()
    errs = [np.abs(res[o] - e).max() / scale for o in (0, 1)]
    mean1 = np.abs(res[1] - e).mean() / scale
    same = np.array_equal(full, res[1])      # the grid size does not change a bit
    ran = not np.array_equal(res[0], res[1])
    ok = np.isfinite(res[1]).all() and errs[1] <= 1.5e-4 and mean1 < 5e-6 and same and ran
    bad += not ok
    print("B%d %3d->%3d %3dx%3d leaky=%d scale %g blocks %d | F(4x4) max err %.2e | F(6x6) max %.2e mean %.2e grid-independent %s ran %s %s"
          % (B, ci, co, h, w, leaky, scale, blocks, errs[0], errs[1], mean1, same, ran, "" if ok else "  <-- BAD"), flush=True)
    if not ok and np.isfinite(res[1]).all():
        d = np.abs(res[1] - e) / scale > 1e-3
        idx = np.argwhere(d)
        if len(idx):
            print("   wrong: %d of %d; imgs" % (len(idx), d.size), sorted(set(idx[:, 0].tolist())), "rows", sorted(set(idx[:, 2].tolist()))[:30], "cols", sorted(set(idx[:, 3].tolist()))[:50], "co", sorted(set(idx[:, 1].tolist()))[:70], flush=True)
print("bad cases", bad)
