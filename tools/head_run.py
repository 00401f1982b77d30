"""GPU box: run the fused head kernel (b2f_op_conv_head16) a few times on a B x 16 x 512 x 960 map, for rocprofv3 (tools/head_sq.sh)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/..")
import numpy as np
from back2future_amd import back2future, ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 6
m = back2future.Model("random:hard:1:1.0")
rng = np.random.default_rng(0)
x = rng.standard_normal((B, 16, 512, 960), dtype=np.float32)
w1 = (rng.standard_normal((16, 16, 3, 3)) / 12).astype(np.float32)
w2 = (rng.standard_normal((32, 16, 3, 3)) / 12).astype(np.float32)
b1 = rng.standard_normal(16).astype(np.float32)
b2 = rng.standard_normal(32).astype(np.float32)
for _ in range(3):
    y = ops.conv_head16(m, x, w1, b1, w2, b2)
print("ok", float(np.abs(y).mean()))
