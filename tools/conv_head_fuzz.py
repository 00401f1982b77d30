"""Random-shape check of the two persistent head kernels (16 -> 16 stride 1, 16 -> 32 stride 2: LDS-DMA double-buffered patch, in-order
s_waitcnt bookkeeping that depends on whether a tile is ragged) against the CPU oracle through the op-level entry point (GPU box):
ragged and tiny maps, many tiles per block, batch > 1, input channel counts below 16.   python tools/conv_head_fuzz.py [seed] [cases]"""
import sys
import numpy as np
sys.path.insert(0, '.')
from back2future_amd import back2future, ops
from oracle import oracle as O

m = back2future.Model("random:hard:1:1.0")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = 0
for it in range(n):
    s2 = bool(rng.integers(2))
    ci = int(rng.choice([16, 16, 16, 9, 12]))
    co = 32 if s2 else 16
    if it % 6 == 5:
        h, w, B = int(rng.integers(100, 400)), int(rng.integers(200, 700)), int(rng.integers(1, 3))     # many tiles per persistent block
    else:
        h, w, B = int(rng.integers(1, 80)), int(rng.integers(1, 150)), int(rng.integers(1, 5))
    x = rng.standard_normal((B, ci, h, w), dtype=np.float32)
    wt = (rng.standard_normal((co, ci, 3, 3), dtype=np.float32) / np.sqrt(9 * ci)).astype(np.float32)
    b = rng.standard_normal(co, dtype=np.float32)
    leaky = bool(rng.integers(2))
    got = ops.conv3x3(m, x, wt, b, 2 if s2 else 1, leaky)
    exp = O.conv3x3(x, wt, b, 2 if s2 else 1, leaky)
    d = float(np.abs(got - exp).max()) if got.shape == exp.shape else float("inf")
    ok = d <= 2e-5
    bad += 0 if ok else 1
    print("%3d %s B%d ci%2d %3dx%3d leaky=%d  max diff %.2e %s" % (it, "16->32 s2" if s2 else "16->16 s1", B, ci, h, w, leaky, d, "" if ok else "FAIL"), flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
