"""Randomized parity sweep of the conv3x3 op entry point against the CPU oracle (GPU box only): shapes that hit every
kernel (direct s1/s2, F(2x2), F(4x4) with full / single N-tile blocks, 16->16, 2-output) and ragged tile edges."""
import sys
import numpy as np
sys.path.insert(0, '.')
from back2future_amd import back2future, ops
from oracle import oracle as O

m = back2future.Model("random:hard:1:1.0")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
worst = 0.0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
for it in range(n):
    stride = int(rng.choice([1, 1, 1, 2]))
    ci = int(rng.choice([3, 8, 16, 24, 32, 40, 64, 96, 100, 128, 196, 200]))
    co = int(rng.choice([2, 7, 16, 20, 32, 36, 64, 68, 96, 100, 128, 160, 192]))
    h, w = int(rng.integers(1, 70)), int(rng.integers(1, 90))
    B = int(rng.integers(1, 4))
    x = rng.standard_normal((B, ci, h, w), dtype=np.float32)
    wt = (rng.standard_normal((co, ci, 3, 3), dtype=np.float32) / np.sqrt(9 * ci)).astype(np.float32)
    b = rng.standard_normal(co, dtype=np.float32)
    leaky = bool(rng.integers(0, 2))
    got = ops.conv3x3(m, x, wt, b, stride, leaky)
    exp = O.conv3x3(x, wt, b, stride, leaky)
    err = float(np.abs(got - exp).max()) if got.size else 0.0
    worst = max(worst, err)
    flag = "" if err < 2e-4 else "   <-- LARGE"
    print("%3d  B%d %3d->%3d s%d %2dx%2d leaky=%d  max err %.2e%s" % (it, B, ci, co, stride, h, w, leaky, err, flag), flush=True)
    assert got.shape == exp.shape and np.isfinite(got).all()
print("worst", worst)
assert worst < 2e-4
