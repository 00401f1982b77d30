"""GPU: the end-to-end contract on weights that behave like trained ones (tests/trained_like.py: every layer rescaled to unit-variance outputs on
the input, plus a variant with 10 x outlier filters) -- for each kernel an integrator can choose for the wide layers: the default (Winograd
F(6x6) on the large maps), F(4x4) (wino6 = 0) and the 1-D Winograd precision option (wino1d = 1).  The real weights cannot be had
(back2future.lua:100-113: Dropbox links), and with plain random weights the activations of the decoders are so small that any rounding
passes.  Prints the table kept in profiles/r06_trained_like_weights.txt."""
import os

import numpy as np
import pytest

from back2future_amd import back2future, flow_io, weights as W
from oracle import oracle as O
from tests import trained_like as TL

pytestmark = pytest.mark.gpu

MEAN = np.array([0.485, 0.456, 0.406] * 3, np.float32).reshape(1, 9, 1, 1)
STD = np.array([0.229, 0.224, 0.225] * 3, np.float32).reshape(1, 9, 1, 1)
KERNELS = (("F(6x6) default", {}), ("F(4x4)", {"wino6": 0}), ("wino1d=1", {"wino6": 0, "wino1d": 1}))


def _check(name, ims, params, past):
    eflow, efo, ebo, fnet, onet = O.compute_flow(ims[0], ims[1], ims[2], params, past, want_net=True)
    m = back2future.Model("random:%s:1:1.0" % ("soft" if past else "hard"))
    rows = []
    try:
        m.set_weights(params)
        for kname, opts in KERNELS:
            with m.options(adaptive_kernels=0, **opts):      # the batch rule (kernel by map size): a single-triplet call would pick per launch
                flow, fo, bo = m.computeFlow(*ims)
            d = np.abs(flow - eflow)
            epe = float(np.sqrt(((flow - eflow) ** 2).sum(0)).mean())
            rows.append((kname, float(d.max()), epe))
            print("%-34s %-16s |flow| max %7.3f  max |dflow| %.2e  EPE %.2e  masks differing %d" % (
                name, kname, float(np.abs(eflow).max()), d.max(), epe, int((fo != efo).sum() + (bo != ebo).sum())), flush=True)
            assert d.max() <= 1e-3 and epe <= 1e-3, (name, kname, d.max(), epe)
    finally:
        m.close()
    return rows


@pytest.mark.parametrize("variant", ["unit-variance", "outlier-filters"])
def test_samples_triplet_with_weights_like_trained(variant):
    """BASELINE.json configs[0]: samples/frame_0009..0011.png, Soft model shape."""
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "samples")
    ims = [flow_io.load_image(os.path.join(d, "frame_%04d.png" % i)) for i in (9, 10, 11)]
    x = np.concatenate(ims, 0)[None]
    xs = O.image_scale_bilinear(((x[0] + (-MEAN[0])) / STD[0]).astype(np.float32), 320, 1216)[None]
    params = W.random_init(7, True, 1.0)
    if variant == "outlier-filters":
        params = TL.add_outliers(params, True)        # first the outliers, then the scale: every layer's output has unit deviation, 3 % of its channels 10 x the rest
    params = TL.calibrate(params, xs, True)
    _check("samples 320x1216 soft " + variant, ims, params, True)


@pytest.mark.parametrize("variant", ["unit-variance", "outlier-filters"])
def test_full_hd_triplet_with_weights_like_trained(variant):
    """One triplet of BASELINE.json configs[4] (3 x 1024 x 1920, Hard model shape)."""
    import torch
    import bench
    x = bench.make_triplets(torch, 1, 1024, 1920, seed=3, device=torch.device("cuda", 0)).cpu().numpy()
    ims = [np.ascontiguousarray(x[0, 3 * f:3 * f + 3]) for f in range(3)]
    xn = ((x + (-MEAN)) / STD).astype(np.float32)
    params = W.random_init(9, False, 1.0)
    if variant == "outlier-filters":
        params = TL.add_outliers(params, False)
    params = TL.calibrate(params, xn, False)
    _check("1024x1920 hard " + variant, ims, params, False)
