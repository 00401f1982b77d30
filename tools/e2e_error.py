"""End-to-end error of the HIP path against the CPU oracle (flow and the whole output table) for both model shapes and for the kernel
choices an integrator can make (default F(4x4) / wino1d = 1 / F(2x2) everywhere); GPU box only.
    python tools/e2e_error.py [gains...]"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from back2future_amd import back2future, weights as W
from oracle import oracle as O

gains = [float(a) for a in sys.argv[1:]] or [1.0, 2.0, 4.0]
H, Wd = 384, 768
r = np.random.default_rng(3)
base = r.random((1, 3, H + 16, Wd + 16), dtype=np.float32)
x = np.concatenate([base[:, :, 8:8 + H, 8:8 + Wd], base[:, :, 7:7 + H, 5:5 + Wd], base[:, :, 6:6 + H, 2:2 + Wd]], 1)
x = ((x - 0.45) / 0.225).astype(np.float32)
for which, past in (("hard", False), ("soft", True)):
    for gain in gains:
        m = back2future.Model("random:%s:7:%s" % (which, gain))
        m.set_option("adaptive_kernels", 0)
        ref = O.pwc_forward(x, m.get_weights(), past)
        row = []
        for name, opts in (("F(4x4) default", {}), ("wino1d=1", {"wino1d": 1}), ("F(2x2) everywhere", {"wino4_min_pixels": 2147483647})):
            for k, v in opts.items():
                m.set_option(k, v)
            outs = m.forward(x)
            errs = [float(np.abs(a - b).max()) for a, b in zip(outs, ref)]
            row.append("%s: flow %.2e table %.2e" % (name, errs[0], max(errs)))
            for k in opts:
                m.set_option(k, {"wino1d": 0, "wino4_min_pixels": 4096}[k])
        print(which, "gain", gain, "(|flow| max %.3g) | " % float(np.abs(ref[0]).max()) + " | ".join(row), flush=True)
        m.close()
