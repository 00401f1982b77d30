"""End-to-end error of the HIP path against the CPU oracle (flow and est[3]) for both model shapes; GPU box only."""
import sys
import numpy as np
sys.path.insert(0, '.')
from back2future_amd import back2future, weights as W
from oracle import oracle as O

for which, past in (("hard", False), ("soft", True)):
    for gain in (1.0, 2.0, 3.0, 4.0):
        m = back2future.Model("random:%s:7:%s" % (which, gain))
        H, Wd = 384, 768
        r = np.random.default_rng(3)
        base = r.random((1, 3, H + 16, Wd + 16), dtype=np.float32)
        x = np.concatenate([base[:, :, 8:8 + H, 8:8 + Wd], base[:, :, 7:7 + H, 5:5 + Wd], base[:, :, 6:6 + H, 2:2 + Wd]], 1)
        x = (x - 0.45) / 0.225
        outs = m.forward(x.astype(np.float32))
        ref = O.pwc_forward(x.astype(np.float32), m.get_weights(), past)
        errs = [float(np.abs(a - b).max()) for a, b in zip(outs, ref)]
        print(which, "gain", gain, "max |flow err| %.3g (|flow| max %.3g)" % (errs[0], float(np.abs(ref[0]).max())),
              " worst over the %d-tensor table %.3g" % (len(errs), max(errs)), flush=True)
