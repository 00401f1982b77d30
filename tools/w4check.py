import sys, numpy as np
sys.path.insert(0, '.')
from back2future_amd import back2future, ops
from oracle import oracle as O
m = back2future.Model("random:hard:1:1.0")
for (ci, co, h, w) in [(64, 64, 16, 32), (8, 64, 4, 4), (96, 96, 9, 30), (192, 192, 4, 7), (196, 128, 16, 33), (128, 128, 40, 70), (32, 96, 37, 19), (64, 70, 18, 34)]:
    r = np.random.default_rng(ci * 1000 + co)
    x = r.standard_normal((2, ci, h, w), dtype=np.float32)
    wt = (r.standard_normal((co, ci, 3, 3), dtype=np.float32) / np.sqrt(9 * ci)).astype(np.float32)
    b = r.standard_normal(co, dtype=np.float32)
    got = ops.conv3x3(m, x, wt, b, 1, True)
    exp = O.conv3x3(x, wt, b, 1, True)
    err = np.abs(got - exp)
    print(ci, co, h, w, "max abs err", err.max(), "at", np.unravel_index(err.argmax(), err.shape), "mean", err.mean(), flush=True)
