"""End-to-end error of the HIP path against the CPU oracle (flow and the whole output table) for both model shapes and for the kernel
choices an integrator can make (default: F(6x6) on the large maps / wino6 = 0: F(4x4) / wino1d = 1 / F(2x2) everywhere); GPU box only.
    python tools/e2e_error.py [gains...]              # 384 x 768, the whole table
    python tools/e2e_error.py --configs               # one triplet at each BASELINE.json size (kernels by map size: the bits of any batch)"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from back2future_amd import back2future, weights as W
from oracle import oracle as O

CHOICES = (("default (F(6x6) >= 16384 px)", {}), ("wino6=0 (F(4x4))", {"wino6": 0}), ("wino6=0 wino1d=1", {"wino6": 0, "wino1d": 1}),
           ("F(2x2) everywhere", {"wino6": 0, "wino4_min_pixels": 2147483647}))
RESET = {"wino6": 1, "wino1d": 0, "wino4_min_pixels": 4096}
if "--configs" in sys.argv:
    import torch
    import bench
    MEAN = np.array([0.485, 0.456, 0.406] * 3, np.float32).reshape(1, 9, 1, 1)
    STD = np.array([0.229, 0.224, 0.225] * 3, np.float32).reshape(1, 9, 1, 1)
    for which, H, Wd in (("hard", 256, 512), ("soft", 384, 1280), ("soft", 448, 1024), ("hard", 1024, 1920), ("soft", 320, 1216)):
        past = which == "soft"
        m = back2future.Model("random:%s:2:1.0" % which)
        m.set_option("adaptive_kernels", 0)
        x = bench.make_triplets(torch, 1, H, Wd, seed=11, device=torch.device("cuda", 0))
        xn = ((x.cpu().numpy() + (-MEAN)) / STD).astype(np.float32)
        ref = O.pwc_forward(xn, W.random_init(2, past, 1.0), past)
        row = []
        for name, opts in CHOICES[:3]:
            for k, v in opts.items():
                m.set_option(k, v)
            outs = m.forward(xn)
            d = outs[0] - ref[0]
            row.append("%s: max %.2e EPE %.2e" % (name, float(np.abs(d).max()), float(np.sqrt((d ** 2).sum(1)).mean())))
            for k in opts:
                m.set_option(k, RESET[k])
        print("%s 3x%dx%d (|flow| max %.3g) | " % (which, H, Wd, float(np.abs(ref[0]).max())) + " | ".join(row), flush=True)
        m.close()
    sys.exit(0)
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
        for name, opts in CHOICES:
            for k, v in opts.items():
                m.set_option(k, v)
            outs = m.forward(x)
            errs = [float(np.abs(a - b).max()) for a, b in zip(outs, ref)]
            row.append("%s: flow %.2e table %.2e" % (name, errs[0], max(errs)))
            for k in opts:
                m.set_option(k, RESET[k])
        print(which, "gain", gain, "(|flow| max %.3g) | " % float(np.abs(ref[0]).max()) + " | ".join(row), flush=True)
        m.close()
