"""Op-level calls through the C ABI, in the layouts of the reference modules
(nn.CostVolMulti, nn.BilinearSamplerBHWD, nn.SpatialConvolution, ...).  Used by the parity
tests; every call runs the HIP kernels of libb2f.so."""
import numpy as np

from . import _lib


def _h(model):
    return model._h


def costvol(model, ref, frm, win=9, fwd=True):
    ref, frm = _lib.f32(ref), _lib.f32(frm)
    B, Cc, h, w = ref.shape
    out = np.empty((B, win * win, h, w), np.float32)
    _lib.check(_lib.lib().b2f_op_costvol(_h(model), _lib.fptr(ref), _lib.fptr(frm), B, Cc, h, w, win, int(bool(fwd)),
                                         _lib.fptr(out)))
    return out


def warp_bhwd(model, img, grid):
    img, grid = _lib.f32(img), _lib.f32(grid)
    B, ih, iw, Cc = img.shape
    _, gh, gw, _two = grid.shape
    out = np.empty((B, gh, gw, Cc), np.float32)
    _lib.check(_lib.lib().b2f_op_warp_bhwd(_h(model), _lib.fptr(img), _lib.fptr(grid), B, ih, iw, Cc, gh, gw,
                                           _lib.fptr(out)))
    return out


def warp_costvol(model, ref, nbr_future, nbr_past, flow, k):
    ref, nf, npast = _lib.f32(ref), _lib.f32(nbr_future), _lib.f32(nbr_past)
    B, Cc, h, w = ref.shape
    fl = _lib.f32(flow) if flow is not None else None
    out = np.empty((B, 162, h, w), np.float32)
    _lib.check(_lib.lib().b2f_op_warp_costvol(_h(model), _lib.fptr(ref), _lib.fptr(nf), _lib.fptr(npast),
                                              _lib.fptr(fl) if fl is not None else None, float(k), B, Cc, h, w,
                                              _lib.fptr(out)))
    return out


def conv3x3(model, x, w, b, stride=1, leaky=False):
    x, w, b = _lib.f32(x), _lib.f32(w), _lib.f32(b)
    B, ci, H, W = x.shape
    co = w.shape[0]
    Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    y = np.empty((B, co, Ho, Wo), np.float32)
    _lib.check(_lib.lib().b2f_op_conv3x3(_h(model), _lib.fptr(x), B, ci, H, W, _lib.fptr(w), _lib.fptr(b), co, stride,
                                         int(bool(leaky)), _lib.fptr(y)))
    return y


def conv_head16(model, x, w1, b1, w2, b2):
    """conv(16,16,s1) + LeakyReLU(0.2) + conv(16,32,s2) + LeakyReLU(0.2) in the fused streaming kernel (pwc.lua:60-62)."""
    x, w1, b1, w2, b2 = _lib.f32(x), _lib.f32(w1), _lib.f32(b1), _lib.f32(w2), _lib.f32(b2)
    B, ci, H, W = x.shape
    assert ci == 16 and w1.shape == (16, 16, 3, 3) and w2.shape == (32, 16, 3, 3)
    y = np.empty((B, 32, (H - 1) // 2 + 1, (W - 1) // 2 + 1), np.float32)
    _lib.check(_lib.lib().b2f_op_conv_head16(_h(model), _lib.fptr(x), B, H, W, _lib.fptr(w1), _lib.fptr(b1), _lib.fptr(w2),
                                             _lib.fptr(b2), _lib.fptr(y)))
    return y


def upsample_flow2x(model, x):
    x = _lib.f32(x)
    B, two, h, w = x.shape
    assert two == 2
    y = np.empty((B, 2, 2 * h, 2 * w), np.float32)
    _lib.check(_lib.lib().b2f_op_upsample_flow2x(_h(model), _lib.fptr(x), B, h, w, _lib.fptr(y)))
    return y


def image_scale(model, src, Hd, Wd, normalize=False):
    """image.scale(src, Wd, Hd) 'bilinear' (back2future.lua:71) on C x Hs x Ws, optionally after ColorNormalize."""
    src = _lib.f32(src)
    Cc, Hs, Ws = src.shape
    dst = np.empty((Cc, Hd, Wd), np.float32)
    _lib.check(_lib.lib().b2f_op_image_scale(_h(model), _lib.fptr(src), Cc, Hs, Ws, int(normalize), _lib.fptr(dst), Hd, Wd))
    return dst


def warp_bhwd_backward(model, img, grid, grad_out, only_grid=False):
    """BilinearSamplerBHWD:updateGradInput: returns (grad_img or None, grad_grid)."""
    img, grid, grad_out = _lib.f32(img), _lib.f32(grid), _lib.f32(grad_out)
    B, ih, iw, Cc = img.shape
    _, gh, gw, _two = grid.shape
    gi = None if only_grid else np.empty_like(img)
    gg = np.empty_like(grid)
    _lib.check(_lib.lib().b2f_op_warp_bhwd_backward(_h(model), _lib.fptr(img), _lib.fptr(grid), _lib.fptr(grad_out), B, ih, iw, Cc,
                                                     gh, gw, _lib.fptr(gi) if gi is not None else None, _lib.fptr(gg)))
    return gi, gg


def costvol_backward(model, ref, frm, grad_out, win=9, fwd=True):
    """CostVolMulti:updateGradInput for {ref, frm}: returns (grad_ref, grad_frm)."""
    ref, frm, grad_out = _lib.f32(ref), _lib.f32(frm), _lib.f32(grad_out)
    B, Cc, h, w = ref.shape
    gr, gf = np.empty_like(ref), np.empty_like(frm)
    _lib.check(_lib.lib().b2f_op_costvol_backward(_h(model), _lib.fptr(ref), _lib.fptr(frm), _lib.fptr(grad_out), B, Cc, h, w, win,
                                                   int(bool(fwd)), _lib.fptr(gr), _lib.fptr(gf)))
    return gr, gf
