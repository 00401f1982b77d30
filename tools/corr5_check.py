"""Bit-identity of the persistent "unit" warp + cost-volume kernel (corr_variant 5) with the two-pixel kernel (variant 3) through
the op-level entry point (GPU box only): ragged maps, every channel count of the pyramid, flows through the border clamp, no
flow, several tiles per block; a third of the cases with a smooth flow (a translation + small noise: the taps of a tile fit the LDS
window of the window-staged forms).   python tools/corr5_check.py [seed] [cases] [variant under test, default 5]"""
import sys
import numpy as np
sys.path.insert(0, '.')
from back2future_amd import back2future, ops

m = back2future.Model("random:hard:1:1.0")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
var = int(sys.argv[3]) if len(sys.argv) > 3 else 5
bad = 0
for it in range(n):
    C = int(rng.choice([32, 64, 96, 128, 192, 16, 48]))
    h, w = int(rng.integers(1, 50)), int(rng.integers(1, 70))
    if it % 5 == 3:
        h, w = int(rng.integers(60, 200)), int(rng.integers(100, 500))     # many tiles per block of the persistent kernel
    B = int(rng.integers(1, 4))
    k = float(rng.choice([0.3125, 0.625, 1.25, 2.5, 5.0]))
    ref = rng.standard_normal((B, C, h, w), dtype=np.float32)
    f3 = rng.standard_normal((B, C, h, w), dtype=np.float32)
    f1 = rng.standard_normal((B, C, h, w), dtype=np.float32)
    noflow = bool(rng.integers(5) == 0)
    flow = None if noflow else (rng.standard_normal((B, 2, h, w)) * float(rng.choice([0.1, 1.0, 8.0]))).astype(np.float32)
    if flow is not None and it % 3 == 1:
        flow = (rng.uniform(-6, 6, (B, 2, 1, 1)) + 0.15 * rng.standard_normal((B, 2, h, w))).astype(np.float32)
    m.set_option("corr_variant", 3)
    a = ops.warp_costvol(m, ref, f3, f1, flow, k)
    m.set_option("corr_variant", var)
    b = ops.warp_costvol(m, ref, f3, f1, flow, k)
    same = np.array_equal(a, b)
    d = float(np.abs(a - b).max())
    bad += 0 if same else 1
    print("%3d B%d C%3d %3dx%3d k=%.4g flow=%s  %s (max diff %.2e, |a|max %.3g)" % (it, B, C, h, w, k, "no" if noflow else "yes",
                                                                                 "bit-identical" if same else "DIFFERENT", d, float(np.abs(a).max())), flush=True)
print("different:", bad)
sys.exit(1 if bad else 0)
