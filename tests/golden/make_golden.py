"""Generates tests/golden/*.npz: small seeded inputs and the outputs of the PyTorch-CPU
witness (tests/torch_ref.py, fp64) for the ops on the computeFlow path and for one whole
model:forward.  Run in the build container (PyTorch is the independent witness of the
Torch7 [3P] semantics); the .npz files are data only and travel to the GPU
box, where both the oracle (-m "not gpu") and the HIP path (-m gpu) are checked against them.

    python tests/golden/make_golden.py

NOTE: the reference itself has no golden vectors and cannot run here (parity unpinned);
these vectors pin the oracle and the kernels to an independent implementation, not to Torch7.
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from back2future_amd import weights as W   # noqa: E402
from tests import torch_ref as R           # noqa: E402


def main():
    rng = np.random.default_rng(20181001)
    t = lambda a: torch.from_numpy(a).double()
    # --- ops ---
    ops = {}
    x = rng.standard_normal((2, 16, 12, 20)).astype(np.float32)
    w = (rng.standard_normal((32, 16, 3, 3)) / 12).astype(np.float32)
    b = rng.standard_normal(32).astype(np.float32)
    ops["conv_x"], ops["conv_w"], ops["conv_b"] = x, w, b
    ops["conv_s1"] = F.leaky_relu(F.conv2d(t(x), t(w), t(b), stride=1, padding=1), 0.2).numpy().astype(np.float32)
    ops["conv_s2"] = F.leaky_relu(F.conv2d(t(x), t(w), t(b), stride=2, padding=1), 0.2).numpy().astype(np.float32)
    ref = rng.standard_normal((2, 32, 10, 14)).astype(np.float32)
    f3 = rng.standard_normal((2, 32, 10, 14)).astype(np.float32)
    f1 = rng.standard_normal((2, 32, 10, 14)).astype(np.float32)
    flow = (rng.standard_normal((2, 2, 10, 14)) * 0.7).astype(np.float32)
    k = 2.5
    ops["cv_ref"], ops["cv_f3"], ops["cv_f1"], ops["cv_flow"], ops["cv_k"] = ref, f3, f1, flow, np.float32(k)
    ops["cv_fwd_nowarp"] = R.costvol_lua(t(ref), t(f3), 9, True).numpy().astype(np.float32)
    ops["cv_bwd_nowarp"] = R.costvol_lua(t(ref), t(f1), 9, False).numpy().astype(np.float32)
    w3 = R.warp_grid_sample(t(f3), t(flow) * k)
    w1 = R.warp_grid_sample(t(f1), t(flow) * -k)
    ops["warp_f3"] = w3.numpy().astype(np.float32)
    ops["cv_joined_warped"] = torch.cat([R.costvol_lua(t(ref), w3, 9, True), R.costvol_lua(t(ref), w1, 9, False)], 1).numpy().astype(np.float32)
    fl = rng.standard_normal((2, 2, 6, 9)).astype(np.float32)
    ops["up_in"] = fl
    ops["up_out"] = F.interpolate(t(fl), scale_factor=2, mode="bilinear", align_corners=True).numpy().astype(np.float32)
    z = (rng.standard_normal((2, 2, 5, 7)) * 3).astype(np.float32)
    ops["sm_in"] = z
    ops["sm_out"] = F.softmax(t(z), 1).numpy().astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "ops.npz"), **ops)

    # --- whole graph, both model kinds, 1 x 9 x 64 x 128, weights = weights.random_init(seed, past, gain) ---
    for past in (False, True):
        xin = rng.standard_normal((1, 9, 64, 128)).astype(np.float32)
        seed, gain = 11, 2.0
        flat = W.random_init(seed, past, gain)
        outs, inter = R.pwc_forward(xin, W.views(flat, past), past)
        d = {"x": xin, "seed": np.int64(seed), "gain": np.float32(gain), "past_flow": np.bool_(past)}
        for i, o in enumerate(outs):
            d["out%02d" % i] = o.astype(np.float32)
        np.savez_compressed(os.path.join(HERE, "forward_%s.npz" % ("soft" if past else "hard")), **d)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
