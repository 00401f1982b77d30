import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from back2future_amd import back2future, ops
from oracle import oracle as O
m = back2future.Model("random:hard:1:1.0")
m.set_option("wino4_min_pixels", 0)
m.set_option("wino2_split", 1)
rng = np.random.default_rng(0)
for mode in ("pattern", "randx", "randw", "rand"):
    for ci in (8, 16, 32, 40, 64):
        h = w = 16
        co = 64
        yy, xx = np.mgrid[0:h, 0:w]
        x = np.zeros((1, ci, h, w), np.float32)
        for c in range(ci):
            x[0, c] = 1 + c + 0.01 * yy + 0.0001 * xx
        wt = np.zeros((co, ci, 3, 3), np.float32)
        for c in range(min(ci, co)):
            wt[c, c, 1, 1] = 1
        if mode in ("randx", "rand"):
            x = rng.standard_normal(x.shape, dtype=np.float32)
        if mode in ("randw", "rand"):
            wt = (rng.standard_normal(wt.shape, dtype=np.float32) / np.sqrt(9 * ci)).astype(np.float32)
        b = np.zeros(co, np.float32)
        y = ops.conv3x3(m, x, wt, b, 1, False)
        e = O.conv3x3(x, wt, b, 1, False)
        bad = ~np.isfinite(y)
        d = np.abs(np.where(bad, 0, y) - e)
        wrong = bad | (d > 1e-3)
        print(mode, "ci", ci, "nan", int(bad.sum()), "wrong", int(wrong.sum()), "of", y.size, "max err", float(d.max()),
              "rows", np.flatnonzero(wrong.any((0, 1, 3))).tolist(), "cols", np.flatnonzero(wrong.any((0, 1, 2))).tolist(),
              "co", np.flatnonzero(wrong.any((0, 2, 3))).tolist()[:12])
