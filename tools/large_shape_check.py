"""Large-shape checks on the GPU box (32-bit offset limits of the buffer-load kernels, arena sizes):
  1. one 2160 x 3840 triplet (net size 2112 x 3840) against the CPU oracle;
  2. a batch of 32 full-HD triplets in ONE forward pass against two passes of 16 (bit-identical) and against
     single passes (equal to rounding: the launcher may choose another Winograd kernel for a single triplet).
    python tools/large_shape_check.py
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from back2future_amd import back2future, weights as W  # noqa: E402
from oracle import oracle as O  # noqa: E402


def triplet(r, H, Wd):
    base = r.random((3, H + 16, Wd + 16), dtype=np.float32)
    for ax in (1, 2):
        base = (base + np.roll(base, 1, ax) + np.roll(base, -1, ax) + np.roll(base, 2, ax) + np.roll(base, -2, ax)) / np.float32(5)
    return [np.ascontiguousarray(base[:, 8 - s:8 - s + H, 8 - 3 * s:8 - 3 * s + Wd]) for s in (0, 1, 2)]


def main():
    r = np.random.default_rng(11)
    m = back2future.Model("random:soft:5:2.0")
    # 1. 4K
    H0, W0 = 2160, 3840
    ims = triplet(r, H0, W0)
    t = time.time()
    flow, fo, bo = m.computeFlow(*ims)
    print("4K computeFlow %.1f ms" % (1e3 * (time.time() - t)), flush=True)
    t = time.time()
    eflow, efo, ebo, fnet, onet = O.compute_flow(*ims, W.random_init(5, True, 2.0), True, want_net=True)
    d = np.abs(flow - eflow)
    epe = np.sqrt(((flow - eflow) ** 2).sum(0)).mean()
    near = O.image_scale_simple((np.abs(onet - 0.6666) < 1e-3).astype(np.uint8), H0, W0).astype(bool)
    bad = int(((fo != efo) & ~near[1:2]).sum() + ((bo != ebo) & ~near[0:1]).sum())
    print("4K vs oracle (%.0f s): max|dflow| %.3g  EPE %.3g  |flow|max %.3g  mask mismatches away from the threshold %d"
          % (time.time() - t, d.max(), epe, np.abs(eflow).max(), bad), flush=True)
    assert d.max() <= 1e-3 and epe <= 1e-3 and bad == 0
    # 2. B = 32 in one pass
    B, H, Wd = 32, 1024, 1920
    x = torch.rand((B, 9, H, Wd), generator=torch.Generator(device="cuda").manual_seed(3), device="cuda")
    flow_b = torch.empty(B, 2, H, Wd, device="cuda"); est3_b = torch.empty(B, 2, H, Wd, device="cuda")
    m.forward_device(x.data_ptr(), B, H, Wd, flow_b.data_ptr(), None, est3_b.data_ptr(), unit_input=True)
    m.synchronize()
    # The same triplets as two passes of 16 and one at a time.  With the default launcher rule the Winograd variant
    # of a layer depends on the number of blocks of the launch, so the passes agree to rounding; with the plain
    # per-map rule (B2F_WINO4_MIN_PIXELS=4096) the same kernels run and not one bit may differ.
    exact = os.environ.get("B2F_WINO4_MIN_PIXELS") is not None
    worst = 0.0
    fh = torch.empty(16, 2, H, Wd, device="cuda"); eh = torch.empty(16, 2, H, Wd, device="cuda")
    for h in (0, 1):
        xh = x[16 * h:16 * h + 16].contiguous()
        m.forward_device(xh.data_ptr(), 16, H, Wd, fh.data_ptr(), None, eh.data_ptr(), unit_input=True)
        m.synchronize()
        worst = max(worst, float((fh - flow_b[16 * h:16 * h + 16]).abs().max()), float((eh - est3_b[16 * h:16 * h + 16]).abs().max()))
    f1 = torch.empty(1, 2, H, Wd, device="cuda"); e1 = torch.empty(1, 2, H, Wd, device="cuda")
    for i in (0, 13, 31):
        xi = x[i:i + 1].contiguous()
        m.forward_device(xi.data_ptr(), 1, H, Wd, f1.data_ptr(), None, e1.data_ptr(), unit_input=True)
        m.synchronize()
        worst = max(worst, float((f1[0] - flow_b[i]).abs().max()), float((e1[0] - est3_b[i]).abs().max()))
    print("B=32 full-HD pass vs 2 x B=16 and single passes: max abs difference %.3g (%s), |flow|max %.3g"
          % (worst, "same kernels: must be 0" if exact else "launcher may pick another Winograd variant", float(flow_b.abs().max())))
    assert worst == 0.0 if exact else worst < 1e-4
    m.close()


if __name__ == "__main__":
    main()
