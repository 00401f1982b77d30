"""GPU: the HIP path against the committed golden vectors (independent of the oracle)."""
import os

import numpy as np
import pytest

from back2future_amd import back2future, ops as K

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def model():
    m = back2future.Model("random:hard:11:2.0")
    yield m
    m.close()


def test_ops_golden(model):
    g = np.load(os.path.join(G, "ops.npz"))
    for s in (1, 2):
        np.testing.assert_allclose(K.conv3x3(model, g["conv_x"], g["conv_w"], g["conv_b"], s, True), g["conv_s%d" % s],
                                   rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(K.costvol(model, g["cv_ref"], g["cv_f3"], 9, True), g["cv_fwd_nowarp"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(K.costvol(model, g["cv_ref"], g["cv_f1"], 9, False), g["cv_bwd_nowarp"], rtol=1e-5, atol=2e-6)
    got = K.warp_costvol(model, g["cv_ref"], g["cv_f3"], g["cv_f1"], g["cv_flow"], float(g["cv_k"]))
    np.testing.assert_allclose(got, g["cv_joined_warped"], rtol=1e-4, atol=5e-6)
    np.testing.assert_allclose(K.upsample_flow2x(model, g["up_in"]), g["up_out"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("kind", ["hard", "soft"])
def test_forward_golden(kind):
    """est[1] (flow) and, for Soft, est[3] (occlusion) / for Hard est[3] (warped image 1) of
    model:forward vs the witness; bar 1e-3 max-abs."""
    import torch
    g = np.load(os.path.join(G, "forward_%s.npz" % kind))
    past = bool(g["past_flow"])
    m = back2future.Model("random:%s:%d:%g" % (kind, int(g["seed"]), float(g["gain"])))
    x = torch.from_numpy(g["x"]).cuda()
    B, _, H, W = x.shape
    flow = torch.zeros(B, 2, H, W, device="cuda")
    occ = torch.zeros(B, 2, H, W, device="cuda")
    est3 = torch.zeros(B, 2 if past else 3, H, W, device="cuda")
    torch.cuda.synchronize()                       # the fills run on torch's stream, the context on its own
    m.forward_device(x.data_ptr(), B, H, W, flow.data_ptr(), occ.data_ptr(), est3.data_ptr())
    m.synchronize()
    assert np.abs(flow.cpu().numpy() - g["out00"]).max() <= 1e-3
    occ_idx = 2 if past else 1                      # output table order, pwc.lua:459-489
    assert np.abs(occ.cpu().numpy() - g["out%02d" % occ_idx]).max() <= 1e-3
    assert np.abs(est3.cpu().numpy() - g["out02"]).max() <= 1e-3
    m.close()


@pytest.mark.parametrize("kind", ["hard", "soft"])
def test_full_output_table_golden(kind):
    """b2f_forward: all 20 / 25 tensors of model:forward (pwc.lua:459-489) vs the witness."""
    g = np.load(os.path.join(G, "forward_%s.npz" % kind))
    m = back2future.Model("random:%s:%d:%g" % (kind, int(g["seed"]), float(g["gain"])))
    outs = m.forward(g["x"])
    assert len(outs) == (25 if kind == "soft" else 20)
    for i, o in enumerate(outs):
        exp = g["out%02d" % i]
        assert o.shape == exp.shape
        assert np.abs(o - exp).max() <= 1e-3, (i, float(np.abs(o - exp).max()))
    m.close()
