"""CPU: the oracle against the committed golden vectors (tests/golden/*.npz, produced by the
PyTorch-CPU witness via tests/golden/make_golden.py).  Runs without PyTorch and without a GPU."""
import os

import numpy as np
import pytest

from back2future_amd import weights as W
from oracle import oracle as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def ops():
    return np.load(os.path.join(G, "ops.npz"))


def test_conv(ops):
    for s in (1, 2):
        got = O.conv3x3(ops["conv_x"], ops["conv_w"], ops["conv_b"], s, leaky=True)
        np.testing.assert_allclose(got, ops["conv_s%d" % s], rtol=1e-5, atol=1e-5)


def test_costvol_and_warp(ops):
    ref, f3, f1, flow, k = ops["cv_ref"], ops["cv_f3"], ops["cv_f1"], ops["cv_flow"], float(ops["cv_k"])
    np.testing.assert_allclose(O.costvol([ref, f3], 9, True), ops["cv_fwd_nowarp"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(O.costvol([ref, f1], 9, False), ops["cv_bwd_nowarp"], rtol=1e-5, atol=1e-6)
    w3 = O.warping_unit(f3, flow, k)
    np.testing.assert_allclose(w3, ops["warp_f3"], rtol=1e-5, atol=1e-5)
    w1 = O.warping_unit(f1, flow, -k)
    joined = np.concatenate([O.costvol([ref, w3], 9, True), O.costvol([ref, w1], 9, False)], 1)
    np.testing.assert_allclose(joined, ops["cv_joined_warped"], rtol=1e-4, atol=5e-6)


def test_upsample_softmax(ops):
    np.testing.assert_allclose(O.upsample_bilinear2x(ops["up_in"]), ops["up_out"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(O.spatial_softmax(ops["sm_in"]), ops["sm_out"], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("kind", ["hard", "soft"])
def test_forward_table(kind):
    g = np.load(os.path.join(G, "forward_%s.npz" % kind))
    past = bool(g["past_flow"])
    flat = W.random_init(int(g["seed"]), past, float(g["gain"]))
    outs = O.pwc_forward(g["x"], flat, past)
    assert len(outs) == (25 if past else 20)
    for i, o in enumerate(outs):
        exp = g["out%02d" % i]
        assert o.shape == exp.shape
        assert np.abs(o - exp).max() <= 1e-3, (i, np.abs(o - exp).max())
    assert np.abs(g["out00"]).max() > 0.05
