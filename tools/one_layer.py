"""One conv layer through the op entry point, a few times (GPU box only; for counter collection on a single kernel).
    python tools/one_layer.py [ci co h w batch reps]"""
import sys
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from back2future_amd import back2future, ops
a = [int(v) for v in sys.argv[1:]] + [128, 128, 128, 240, 4, 3][len(sys.argv) - 1:]
ci, co, h, w, B, reps = a
m = back2future.Model("random:hard:1:1.0")
import os
for kv in os.environ.get("B2F_ONE_LAYER_OPTS", "").split(","):
    if "=" in kv:
        m.set_option(kv.split("=")[0], int(kv.split("=")[1]))
rng = np.random.default_rng(0)
x = rng.standard_normal((B, ci, h, w), dtype=np.float32)
wt = (rng.standard_normal((co, ci, 3, 3), dtype=np.float32) / np.sqrt(9 * ci)).astype(np.float32)
b = rng.standard_normal(co, dtype=np.float32)
for _ in range(reps):
    y = ops.conv3x3(m, x, wt, b, 1, True)
print("ok", float(np.abs(y).mean()))
