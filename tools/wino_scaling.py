#!/usr/bin/env python3
"""Per-chunk cost vs fixed per-block cost of the Winograd conv kernel: times conv3x3 (op-level,
kernel time via rocprofv3 --kernel-trace) for several Cin at a fixed output size."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from back2future_amd import back2future, ops
m = back2future.Model("random:hard")
rng = np.random.default_rng(0)
for ci in (32, 64, 128, 256):
    for co in (128, 64, 32):
        x = rng.standard_normal((2, ci, 256, 480), dtype=np.float32)
        w = (rng.standard_normal((co, ci, 3, 3), dtype=np.float32) / np.sqrt(9 * ci)).astype(np.float32)
        b = np.zeros(co, np.float32)
        ops.conv3x3(m, x, w, b, 1, True)
m.close()
