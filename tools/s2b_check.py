"""GPU box: the stride-2 loader / consumer kernel on the bf16 pipe (option s2_loader = 1, b2f_s2b.hip) against an fp64 convolution, the
fp32-MFMA direct kernel (bf16_conv = 0) and conv3x3_bf6 (s2_loader = 0).
    python tools/s2b_check.py [seed] [nrandom]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from back2future_amd import back2future, ops

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
nrand = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rng = np.random.default_rng(seed)
m = back2future.Model("random:hard:1:1.0")
bad = 0
cases = [(1, 32, 64, 16, 64, 1.0, 1), (2, 32, 64, 37, 71, 1.0, 1), (1, 64, 96, 33, 50, 20.0, 1), (3, 96, 128, 9, 130, 1e-2, 1), (1, 128, 192, 32, 60, 1.0, 1),
         (2, 40, 64, 20, 20, 1.0, 3), (1, 64, 100, 31, 33, 1.0, 1), (1, 24, 32, 40, 66, 1.0, 2), (1, 8, 256, 2, 2, 1.0, 1), (4, 16, 36, 1, 1, 1.0, 1), (3, 72, 160, 64, 48, 1.0, 5)]
for _ in range(nrand):
    cases.append((int(rng.integers(1, 4)), int(rng.integers(1, 20)) * 8, int(rng.integers(2, 65)) * 4, int(rng.integers(1, 70)), int(rng.integers(1, 140)),
                  float(10.0 ** rng.integers(-3, 3)), int(rng.choice([1, 1, 2, 7]))))
for (B, ci, co, h, w, scale, blocks) in cases:
    x = (rng.standard_normal((B, ci, h, w)) * scale).astype(np.float32)
    wt = (rng.standard_normal((co, ci, 3, 3)) / np.sqrt(9 * ci)).astype(np.float32)
    b = (rng.standard_normal(co) * scale).astype(np.float32)
    leaky = bool(rng.integers(2))
    m.set_option("bf16_conv", 0)
    f32 = ops.conv3x3(m, x, wt, b, 2, leaky)
    m.set_option("bf16_conv", 1)
    m.set_option("s2_loader", 0)
    bf6 = ops.conv3x3(m, x, wt, b, 2, leaky)
    m.set_option("s2_loader", 2)
    m.set_option("wino4_persistent", blocks)
    got = ops.conv3x3(m, x, wt, b, 2, leaky)
    m.set_option("wino4_persistent", 1)
    full = ops.conv3x3(m, x, wt, b, 2, leaky)
    y = torch.nn.functional.conv2d(torch.from_numpy(x).double(), torch.from_numpy(wt).double(), torch.from_numpy(b).double(), padding=1, stride=2)
    e = (torch.where(y > 0, y, 0.2 * y) if leaky else y).numpy()
    errs = [np.abs(r - e).max() / scale for r in (f32, bf6, got)]
    same = np.array_equal(full, got)
    ok = np.isfinite(got).all() and errs[2] < 4 * max(errs[0], 2e-6) and same
    bad += not ok
    print("B%d %3d->%3d %3dx%3d leaky=%d scale %g blocks %d | fp32 kernel max err %.2e | bf6 %.2e | loader/consumer %.2e grid-independent %s %s"
          % (B, ci, co, h, w, leaky, scale, blocks, errs[0], errs[1], errs[2], same, "" if ok else "  <-- BAD"))
    if not ok and np.isfinite(got).all():
        idx = np.argwhere(np.abs(got - e) / scale > 1e-4)
        if len(idx):
            print("   wrong: imgs", sorted(set(idx[:, 0].tolist())), "rows", sorted(set(idx[:, 2].tolist()))[:30], "cols", sorted(set(idx[:, 3].tolist()))[:40], "co", sorted(set(idx[:, 1].tolist()))[:33])
print("bad cases", bad)
assert bad == 0
