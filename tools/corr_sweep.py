"""Randomized parity sweep of the fused warp + cost-volume op against the CPU oracle (GPU box only): channel counts,
ragged map sizes, flow scales incl. far out-of-range displacements, both kernel instantiations."""
import os
import sys
import numpy as np
sys.path.insert(0, '.')
from back2future_amd import back2future, ops
from oracle import oracle as O

m = back2future.Model("random:hard:1:1.0")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20      # the oracle side takes ~9 s per case
worst = 0.0
for it in range(n):
    C = int(rng.choice([8, 16, 32, 64, 96, 128, 192]))
    h, w = int(rng.integers(1, 50)), int(rng.integers(1, 70))
    if it % 7 == 3:
        h, w = int(rng.integers(60, 140)), int(rng.integers(100, 300))     # several tiles per block of the persistent kernel
    B = int(rng.integers(1, 4))
    k = float(rng.choice([0.3125, 0.625, 1.25, 2.5, 5.0]))
    var = int(rng.choice([-1, 0, 1, 3, 5, 7]))   # the product library's instantiations (-1: the launcher's choice)
    m.set_option("corr_variant", var)
    ref = rng.standard_normal((B, C, h, w), dtype=np.float32)
    f3 = rng.standard_normal((B, C, h, w), dtype=np.float32)
    f1 = rng.standard_normal((B, C, h, w), dtype=np.float32)
    noflow = bool(rng.integers(5) == 0)
    flow = None if noflow else (rng.standard_normal((B, 2, h, w)) * float(rng.choice([0.1, 1.0, 8.0]))).astype(np.float32)
    got = ops.warp_costvol(m, ref, f3, f1, flow, k)
    if noflow:
        exp = np.concatenate([O.costvol([ref, f3], 9, True), O.costvol([ref, f1], 9, False)], 1)
    else:
        exp = np.concatenate([O.costvol([ref, O.warping_unit(f3, flow, k)], 9, True), O.costvol([ref, O.warping_unit(f1, flow, -k)], 9, False)], 1)
    err = float(np.abs(got - exp).max())
    worst = max(worst, err)
    print("%3d B%d C%3d %2dx%2d k=%.4g variant=%s flow=%s  max err %.2e%s" % (it, B, C, h, w, k, var, "no" if noflow else "yes", err,
                                                                     "" if err < 2e-5 else "   <-- LARGE"), flush=True)
    assert np.isfinite(got).all() and err < 1e-4
print("worst", worst)
