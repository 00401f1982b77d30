"""Persistent F(4x4) kernel (conv3x3_wino4p) against the one-tile-per-block kernel and the CPU oracle through the conv3x3 op
entry point (GPU box only): random shapes with ragged edges, odd / even chunk counts, one to many tiles per block."""
import sys
import numpy as np
sys.path.insert(0, '.')
from back2future_amd import back2future, ops
from oracle import oracle as O

m = back2future.Model("random:hard:1:1.0")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
worst_o, worst_k = 0.0, 0.0
for it in range(n):
    ci = int(rng.choice([8, 16, 24, 32, 40, 64, 96, 104, 128, 200]))
    co = int(rng.choice([32, 64, 68, 96, 100, 128, 160, 192]))
    h, w = int(rng.integers(1, 70)), int(rng.integers(1, 100))
    B = int(rng.integers(1, 4))
    x = rng.standard_normal((B, ci, h, w), dtype=np.float32)
    wt = (rng.standard_normal((co, ci, 3, 3), dtype=np.float32) / np.sqrt(9 * ci)).astype(np.float32)
    b = rng.standard_normal(co, dtype=np.float32)
    leaky = bool(rng.integers(0, 2))
    m.set_option("wino4_min_pixels", 0)          # F(4x4) at every size
    m.set_option("wino4_persistent", 0)
    ref = ops.conv3x3(m, x, wt, b, 1, leaky)
    G = int(rng.choice([2, 3, 5, 8, 17, 64]))
    m.set_option("wino4_persistent", G)
    got = ops.conv3x3(m, x, wt, b, 1, leaky)
    exp = O.conv3x3(x, wt, b, 1, leaky)
    ek = float(np.abs(got - ref).max()) if got.size else 0.0
    eo = float(np.abs(got - exp).max()) if got.size else 0.0
    worst_k, worst_o = max(worst_k, ek), max(worst_o, eo)
    flag = "" if (ek == 0 and eo < 2e-4) else "   <-- DIFFERENT"
    print("%3d  B%d %3d->%3d %2dx%2d leaky=%d G=%2d  vs one-tile kernel %.2e  vs oracle %.2e%s" % (it, B, ci, co, h, w, leaky, G, ek, eo, flag), flush=True)
    assert got.shape == exp.shape and np.isfinite(got).all()
print("worst vs one-tile kernel %.3g, vs oracle %.3g" % (worst_k, worst_o))
assert worst_k == 0 and worst_o < 2e-4
