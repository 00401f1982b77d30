"""CPU: the oracle's backward restatements checked the way the reference checks its own natives --
extras/stnbhwd/test.lua:95-118 runs nn.Jacobian.testJacobian (finite differences against the analytic gradient) on the
sampler for images and grids -- plus torch.autograd as an independent witness."""
import numpy as np
import torch

from oracle import oracle as O
from tests import torch_ref as R


def _num_grad(f, x, go, eps=1e-3):
    """d <f(x), go> / dx by central differences (float64 accumulation of the float32 forward)."""
    g = np.zeros_like(x, dtype=np.float64)
    it = np.nditer(x, flags=["multi_index"])
    for _ in it:
        i = it.multi_index
        old = x[i]
        x[i] = old + eps; a = float((f(x).astype(np.float64) * go).sum())
        x[i] = old - eps; b = float((f(x).astype(np.float64) * go).sum())
        x[i] = old
        g[i] = (a - b) / (2 * eps)
    return g


def test_sampler_backward_jacobian():
    rng = np.random.default_rng(0)
    B, h, w, C = 1, 5, 6, 3
    img = rng.standard_normal((B, h, w, C)).astype(np.float32)
    # keep the sampling points away from integer coordinates (the interpolant has kinks there) and inside the image
    grid = (rng.uniform(0.2, 0.8, (B, h, w, 2)) * rng.choice([-1, 1], (B, h, w, 2))).astype(np.float32)
    xs, ys = np.meshgrid(np.arange(w), np.arange(h))
    grid[..., 0] = np.clip(xs + grid[..., 0], 0.2, w - 1.2) - xs
    grid[..., 1] = np.clip(ys + grid[..., 1], 0.2, h - 1.2) - ys
    go = rng.standard_normal((B, h, w, C)).astype(np.float32)
    gi, gg = O.warp_bhwd_backward(img, grid, go)
    ni = _num_grad(lambda a: O.warp_bhwd(a, grid), img.copy(), go)
    ng = _num_grad(lambda a: O.warp_bhwd(img, a), grid.copy(), go, eps=2e-3)
    np.testing.assert_allclose(gi, ni, rtol=0, atol=2e-3)
    np.testing.assert_allclose(gg, ng, rtol=0, atol=5e-3)
    gi2, gg2 = O.warp_bhwd_backward(img, grid, go, only_grid=True)      # the onlyGrid instantiation
    assert gi2 is None
    np.testing.assert_array_equal(gg2, gg)


def test_sampler_backward_vs_autograd():
    rng = np.random.default_rng(1)
    B, h, w, C = 2, 7, 9, 40          # > 32 channels: the strided partial sums of the CUDA kernel wrap around
    img = torch.from_numpy(rng.standard_normal((B, C, h, w))).double().requires_grad_(True)
    flow = torch.from_numpy(rng.uniform(-2.5, 2.5, (B, 2, h, w))).double().requires_grad_(True)
    out = R.warp_gather(img, flow)
    go = torch.from_numpy(rng.standard_normal((B, C, h, w))).double()
    out.backward(go)
    gi, gg = O.warp_bhwd_backward(img.detach().permute(0, 2, 3, 1).float().numpy(), flow.detach().permute(0, 2, 3, 1).float().numpy(),
                                  go.permute(0, 2, 3, 1).float().numpy())
    np.testing.assert_allclose(gi, img.grad.permute(0, 2, 3, 1).numpy(), rtol=0, atol=1e-5)
    # autograd sees the clamp (zero gradient where the coordinate was clamped); the CUDA kernel does not (BilinearSamplerBHWD.cu:289-290)
    x = torch.arange(w)[None, None, :] + flow.detach()[:, 0]
    y = torch.arange(h)[None, :, None] + flow.detach()[:, 1]
    inside = ((x > 0) & (x < w - 1) & (y > 0) & (y < h - 1)).numpy()
    ref = flow.grad.permute(0, 2, 3, 1).numpy()
    assert inside.sum() > 20
    np.testing.assert_allclose(gg[inside], ref[inside], rtol=0, atol=2e-5)


def test_costvol_backward_vs_autograd():
    rng = np.random.default_rng(2)
    for win, fwd in [(9, True), (9, False), (5, True), (3, False)]:
        ref = torch.from_numpy(rng.standard_normal((2, 6, 10, 12))).double().requires_grad_(True)
        frm = torch.from_numpy(rng.standard_normal((2, 6, 10, 12))).double().requires_grad_(True)
        out = R.costvol_lua(ref, frm, win, fwd)
        go = torch.from_numpy(rng.standard_normal(tuple(out.shape))).double()
        out.backward(go)
        gr, gf = O.costvol_backward(ref.detach().float().numpy(), frm.detach().float().numpy(), go.float().numpy(), win, fwd)
        np.testing.assert_allclose(gr, ref.grad.numpy(), rtol=0, atol=1e-5)
        np.testing.assert_allclose(gf, frm.grad.numpy(), rtol=0, atol=1e-5)
