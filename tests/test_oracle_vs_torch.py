"""CPU: the C oracle against PyTorch-CPU as an independent witness, op by op and
end to end, plus known-answer tests derived from the reference text."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from back2future_amd import weights as W
from oracle import oracle as O
from tests import torch_ref as R

torch.set_num_threads(4)


def test_param_counts():
    # 7 193 316 (Hard) / 10 168 302 (Soft), from pwc.lua:58-85
    assert O.param_count(False) == 7193316 == W.param_count(False)
    assert O.param_count(True) == 10168302 == W.param_count(True)


@pytest.mark.parametrize("ci,co,stride,h,w", [(3, 16, 2, 12, 20), (16, 16, 1, 7, 9), (5, 7, 2, 9, 11), (32, 2, 1, 6, 5)])
def test_conv3x3(rng, ci, co, stride, h, w):
    x = rng.standard_normal((2, ci, h, w), dtype=np.float32)
    wt = rng.standard_normal((co, ci, 3, 3), dtype=np.float32) * 0.2
    b = rng.standard_normal(co, dtype=np.float32)
    y = O.conv3x3(x, wt, b, stride, leaky=True)
    ref = F.leaky_relu(F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(wt).double(),
                                torch.from_numpy(b).double(), stride=stride, padding=1), 0.2).numpy()
    assert y.shape == ref.shape
    np.testing.assert_allclose(y, ref, rtol=1e-5, atol=1e-5)


def test_pool_upsample_softmax(rng):
    x = rng.standard_normal((2, 3, 8, 12), dtype=np.float32)
    t = torch.from_numpy(x)
    np.testing.assert_allclose(O.avgpool2(x), F.avg_pool2d(t, 2).numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(O.upsample_bilinear2x(x),
                               F.interpolate(t, scale_factor=2, mode="bilinear", align_corners=True).numpy(),
                               rtol=1e-5, atol=1e-6)
    np.testing.assert_array_equal(O.upsample_nearest2x(x), F.interpolate(t, scale_factor=2, mode="nearest").numpy())
    z = rng.standard_normal((2, 2, 5, 7), dtype=np.float32) * 4
    np.testing.assert_allclose(O.spatial_softmax(z), F.softmax(torch.from_numpy(z), 1).numpy(), rtol=1e-6, atol=1e-7)
    # 1-pixel-high map (level 7 of a 64-row input)
    x1 = rng.standard_normal((1, 2, 1, 3), dtype=np.float32)
    np.testing.assert_allclose(O.upsample_bilinear2x(x1),
                               F.interpolate(torch.from_numpy(x1), scale_factor=2, mode="bilinear", align_corners=True).numpy(),
                               rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("win,fwd", [(9, True), (9, False), (5, True), (5, False)])
def test_costvol_vs_lua_transcription(rng, win, fwd):
    ref = rng.standard_normal((2, 6, 11, 13), dtype=np.float32)
    frm = rng.standard_normal((2, 6, 11, 13), dtype=np.float32)
    got = O.costvol([ref, frm], win, fwd)
    exp = R.costvol_lua(torch.from_numpy(ref).double(), torch.from_numpy(frm).double(), win, fwd).numpy()
    np.testing.assert_allclose(got, exp, rtol=1e-5, atol=1e-6)
    # closed form of SURVEY s8 a5: channel c=(qx+n)*win+(qy+n); fwd pairs ref(y,x) with frm(y-qy,x-qx)
    n = (win - 1) // 2
    s = 1 if fwd else -1
    for (qx, qy) in [(-n, -n), (0, 0), (n, -1), (2, n)]:
        c = (qx + n) * win + (qy + n)
        y, x = 6, 7
        yy, xx = y - s * qy, x - s * qx
        v = (ref[:, :, y, x] * frm[:, :, yy, xx]).sum(1) / 6 if (0 <= yy < 11 and 0 <= xx < 13) else 0
        np.testing.assert_allclose(got[:, c, y, x], v, rtol=1e-5, atol=1e-6)


def test_costvol_impulse_known_answer():
    """CostVolMulti.lua:225-254: a point moving by (+1,+1) per frame lights the SAME
    channel c = (-1+n)*win + (-1+n) in the fwd volume (ref vs next) and the bwd volume
    (ref vs previous)."""
    win, n = 9, 4
    h = w = 12
    prev = np.zeros((1, 1, h, w), np.float32); cur = prev.copy(); nxt = prev.copy()
    prev[0, 0, 4, 5] = 1; cur[0, 0, 5, 6] = 1; nxt[0, 0, 6, 7] = 1
    f = O.costvol([cur, nxt], win, True)
    b = O.costvol([cur, prev], win, False)
    c = (-1 + n) * win + (-1 + n)
    assert f[0, c, 5, 6] == 1 and b[0, c, 5, 6] == 1
    assert f.sum() == 1 and b.sum() == 1


def test_costvol_uniform_known_answer():
    """uniform features: cost = mean(ref^2) inside, 0 in the out-of-range bands."""
    x = np.full((1, 4, 10, 10), 0.5, np.float32)
    f = O.costvol([x, x], 9, True)
    c = (4 + 4) * 9 + (0 + 4)            # qx=+4, qy=0: valid x >= 4
    assert np.allclose(f[0, c, :, 4:], 0.25) and np.all(f[0, c, :, :4] == 0)


def test_warp_vs_grid_sample(rng):
    B, C, h, w = 2, 5, 9, 12
    img = rng.standard_normal((B, C, h, w), dtype=np.float32)
    flow = (rng.standard_normal((B, 2, h, w)) * 3).astype(np.float32)
    flow[0, :, 0, 0] = (-50, -50); flow[0, :, 1, 1] = (50, 50); flow[0, :, 2, 2] = (0.0, 0.0)
    got = O.warping_unit(img, flow, 1.0)
    exp = R.warp_grid_sample(torch.from_numpy(img), torch.from_numpy(flow)).numpy()
    np.testing.assert_allclose(got, exp, rtol=1e-5, atol=1e-5)
    exp2 = R.warp_gather(torch.from_numpy(img), torch.from_numpy(flow)).numpy()
    np.testing.assert_allclose(got, exp2, rtol=1e-5, atol=1e-5)


def test_warp_known_answers(rng):
    img = rng.standard_normal((1, 4, 6, 7, ), dtype=np.float32)       # BHWD: 1 x 4 x 6 x 7?  -> use B,H,W,D
    img = rng.standard_normal((1, 6, 7, 3), dtype=np.float32)
    zero = np.zeros((1, 6, 7, 2), np.float32)
    np.testing.assert_array_equal(O.warp_bhwd(img, zero), img)          # zero flow = identity
    far = np.zeros((1, 6, 7, 2), np.float32); far[..., 0] = 100         # beyond right border -> edge replicate
    out = O.warp_bhwd(img, far)
    np.testing.assert_array_equal(out, np.broadcast_to(img[:, :, -1:, :], img.shape))
    one = np.zeros((1, 6, 7, 2), np.float32); one[..., 1] = -1          # integer shift up by one row
    out = O.warp_bhwd(img, one)
    np.testing.assert_array_equal(out[:, 1:], img[:, :-1])
    np.testing.assert_array_equal(out[:, 0], img[:, 0])
    # the grid sizes the output (BilinearSamplerBHWD.lua:70)
    g = np.zeros((1, 3, 4, 2), np.float32)
    assert O.warp_bhwd(img, g).shape == (1, 3, 4, 3)


def test_image_scale(rng):
    src = rng.random((3, 10, 17), dtype=np.float32)
    np.testing.assert_array_equal(O.image_scale_bilinear(src, 10, 17), src)        # equal size = copy
    up = O.image_scale_bilinear(src, 19, 33)                                         # exact x2-1: align-corners lerp
    ref = F.interpolate(torch.from_numpy(src)[None], size=(19, 33), mode="bilinear", align_corners=True)[0].numpy()
    np.testing.assert_allclose(up, ref, rtol=1e-5, atol=1e-6)
    dn = O.image_scale_bilinear(src, 5, 17)                                          # integer ratio 2 in y: box mean
    np.testing.assert_allclose(dn, 0.5 * (src[:, 0::2] + src[:, 1::2]), rtol=1e-6, atol=1e-6)
    # mean preservation for a fractional box filter (375->320 style ratio)
    big = rng.random((1, 75, 90), dtype=np.float32)
    small = O.image_scale_bilinear(big, 64, 64)
    assert abs(small.mean() - big.mean()) < 2e-3
    d = rng.random((2, 8, 8))
    s = O.image_scale_simple(d, 11, 13)
    jj = np.minimum((np.arange(11, dtype=np.float32) * np.float32(8 / 11)).astype(np.int64), 7)
    ii = np.minimum((np.arange(13, dtype=np.float32) * np.float32(8 / 13)).astype(np.int64), 7)
    np.testing.assert_array_equal(s, d[:, jj][:, :, ii])


def _area_matrix(s, d):
    """d x s matrix of the exact box filter for down-scaling a line of s samples to d: output i averages the piecewise-
    constant signal over [i s/d, (i+1) s/d) -- overlap lengths in float64, no running accumulator."""
    scale = s / d
    m = np.zeros((d, s), np.float64)
    for i in range(d):
        lo, hi = i * scale, (i + 1) * scale
        for j in range(int(np.floor(lo)), min(int(np.ceil(hi)), s)):
            m[i, j] = max(0.0, min(hi, j + 1) - max(lo, j))
        m[i] /= m[i].sum()
    return m


def _lerp_matrix(s, d):
    """d x s matrix of image.scale's up-scaling: align-corners linear interpolation, last sample copied."""
    m = np.zeros((d, s), np.float64)
    for i in range(d - 1):
        t = i * (s - 1) / (d - 1)
        j = int(np.floor(t))
        m[i, j] += 1 - (t - j)
        if t - j > 0:
            m[i, j + 1] += t - j
    m[d - 1, s - 1] = 1
    return m


@pytest.mark.parametrize("Hs,Ws,Hd,Wd", [(375, 1242, 320, 1216), (436, 1024, 384, 1024), (100, 37, 64, 64), (64, 200, 64, 128),
                                         (70, 70, 64, 128)])
def test_image_scale_area_integral_witness(rng, Hs, Ws, Hd, Wd):
    """image.scale 'bilinear' (back2future.lua:71) against an independent float64 witness: down-scaling is the exact
    area integral of the piecewise-constant image (brute-force overlap lengths, no running accumulator), up-scaling the
    align-corners lerp, width pass first.  Tolerance: the routine carries its positions, weights and accumulators in
    float (torch/image generic/image.c [3P]); a position near 1200 is rounded to 1.2e-4 pixel, which moves a box edge by
    that much: |error| <= 1.2e-4 x (neighbour difference <= 1) on [0, 1] data, 1e-4 with the 375 x 1242 case measured
    at 2.4e-5; small images (positions below 256) 1e-5."""
    src = rng.random((2, Hs, Ws), dtype=np.float32)
    mw = _area_matrix(Ws, Wd) if Wd < Ws else (_lerp_matrix(Ws, Wd) if Wd > Ws else np.eye(Ws))
    mh = _area_matrix(Hs, Hd) if Hd < Hs else (_lerp_matrix(Hs, Hd) if Hd > Hs else np.eye(Hs))
    ref = np.einsum("ij,cjk->cik", mh, np.einsum("cjk,lk->cjl", src.astype(np.float64), mw))
    got = O.image_scale_bilinear(src, Hd, Wd)
    assert got.shape == ref.shape
    tol = 1e-4 if max(Hs, Ws) > 256 else 1e-5
    assert np.abs(got - ref).max() <= tol, float(np.abs(got - ref).max())
    assert np.abs(got - ref).mean() <= 1e-5


@pytest.mark.parametrize("s,d", [(375, 320), (1242, 1216), (7, 3), (64, 64), (5, 17), (320, 375), (1216, 1242)])
def test_image_scale_simple_witness(s, d):
    """image.scale 'simple' (back2future.lua:82,89,91): nearest, source index = floor(dst * s / d) computed in float
    [3P]; the witness computes the same index in exact rational arithmetic -- the two may only differ where the float
    product lands on the other side of an integer (never for these sizes; the assertion documents it)."""
    from fractions import Fraction
    src = np.arange(s, dtype=np.float64)[None, None, :].repeat(2, 1)
    got = O.image_scale_simple(src, 2, d)[0, 0].astype(np.int64)
    exact = np.array([min(int(Fraction(i * s, d)), s - 1) for i in range(d)], np.int64)
    np.testing.assert_array_equal(got, exact)


def test_color_normalize():
    x = np.full((9, 2, 2), 0.5, np.float32)
    y = O.color_normalize(x)
    mean = np.array([0.485, 0.456, 0.406], np.float32); std = np.array([0.229, 0.224, 0.225], np.float32)
    for c in range(9):
        assert np.allclose(y[c], (np.float32(0.5) - mean[c % 3]) / std[c % 3], rtol=1e-6)


@pytest.mark.parametrize("past_flow", [False, True])
def test_pwc_forward_end_to_end(past_flow):
    """Whole graph (all 20 / 25 outputs) at 128 x 192, oracle (fp32 C) vs torch (fp64)."""
    H, Wd = 128, 192
    rng = np.random.default_rng(7)
    x = rng.standard_normal((1, 9, H, Wd)).astype(np.float32)
    flat = W.random_init(seed=5, past_flow=past_flow, gain=2.0)
    outs = O.pwc_forward(x, flat, past_flow)
    ref, inter = R.pwc_forward(x, W.views(flat, past_flow), past_flow)
    assert len(outs) == len(ref) == (25 if past_flow else 20)
    assert np.abs(ref[0]).max() > 0.05        # flows are big enough for the warps to matter
    for i, (a, b) in enumerate(zip(outs, ref)):
        assert a.shape == b.shape
        assert np.abs(a - b).max() < 1e-3, (i, np.abs(a - b).max())
