"""GPU box: the direct conv kernel on the bf16 pipe (option bf16_conv = 1, b2f_convb.hip) against fp64 and the fp32-MFMA direct kernel.
    python tools/convb_check.py [seed]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
os.environ.setdefault("B2F_WINO", "1")
from back2future_amd import back2future, ops

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rng = np.random.default_rng(seed)
m = back2future.Model("random:hard:1:1.0")
bad = 0
cases = [(1, 32, 64, 16, 64, 2, 1.0), (2, 32, 64, 37, 71, 2, 1.0), (1, 64, 96, 33, 50, 2, 20.0), (3, 96, 128, 9, 130, 2, 1e-2), (1, 128, 192, 32, 60, 2, 1.0),
         (2, 40, 64, 20, 20, 2, 1.0), (1, 64, 100, 31, 33, 2, 1.0), (1, 24, 32, 40, 66, 2, 1.0)]
for (B, ci, co, h, w, stride, scale) in cases:
    x = (rng.standard_normal((B, ci, h, w)) * scale).astype(np.float32)
    wt = (rng.standard_normal((co, ci, 3, 3)) / np.sqrt(9 * ci)).astype(np.float32)
    b = (rng.standard_normal(co) * scale).astype(np.float32)
    leaky = bool(rng.integers(2))
    res = {}
    for opt in (0, 1):
        m.set_option("bf16_conv", opt)
        res[opt] = ops.conv3x3(m, x, wt, b, stride, leaky)
    y = torch.nn.functional.conv2d(torch.from_numpy(x).double(), torch.from_numpy(wt).double(), torch.from_numpy(b).double(), padding=1, stride=stride)
    e = (torch.where(y > 0, y, 0.2 * y) if leaky else y).numpy()
    errs = [np.abs(res[o] - e).max() / scale for o in (0, 1)]
    ok = np.isfinite(res[1]).all() and errs[1] < 4 * max(errs[0], 2e-6) and not np.array_equal(res[0], res[1])
    bad += not ok
    print("B%d %d->%d %dx%d s%d leaky=%d scale %g | fp32 kernel max err %.2e | bf16x6 %.2e %s" % (B, ci, co, h, w, stride, leaky, scale, errs[0], errs[1], "" if ok else "  <-- BAD"))
    if not ok and np.isfinite(res[1]).all():
        d = np.abs(res[1] - e) / scale > 1e-4
        idx = np.argwhere(d)
        if len(idx):
            print("   wrong: rows", sorted(set(idx[:, 2].tolist()))[:30], "cols", sorted(set(idx[:, 3].tolist()))[:40], "co", sorted(set(idx[:, 1].tolist()))[:33])
print("bad cases", bad)
assert bad == 0
