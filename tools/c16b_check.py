"""GPU box: the 16 -> 16 head layer on the bf16 pipe (bf16_direct = 1) against fp64 and against the fp32-MFMA kernel.
    python tools/c16b_check.py [seed]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from back2future_amd import back2future, ops

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rng = np.random.default_rng(seed)
m = back2future.Model("random:hard:1:1.0")
bad = 0
for (B, ci, co, h, w, stride, scale) in [(1, 16, 16, 16, 32, 1, 1.0), (2, 16, 16, 37, 70, 1, 1.0), (1, 16, 16, 5, 9, 1, 50.0), (3, 16, 16, 64, 33, 1, 1e-3), (1, 12, 16, 40, 40, 1, 1.0),
                                       (1, 16, 32, 32, 64, 2, 1.0), (2, 16, 32, 37, 71, 2, 1.0), (1, 16, 32, 9, 11, 2, 30.0)]:
    x = (rng.standard_normal((B, ci, h, w)) * scale).astype(np.float32)
    wt = (rng.standard_normal((co, ci, 3, 3)) / np.sqrt(9 * ci)).astype(np.float32)
    b = rng.standard_normal(co).astype(np.float32) * np.float32(scale)
    leaky = bool(rng.integers(2))
    res = {}
    for opt in (0, 1):
        m.set_option("bf16_direct", opt)
        res[opt] = ops.conv3x3(m, x, wt, b, stride, leaky)
    # fp64 reference
    xp = np.pad(x.astype(np.float64), ((0, 0), (0, 0), (1, 1), (1, 1)))
    ho, wo = res[0].shape[2:]
    e = np.zeros((B, co, ho, wo))
    for ky in range(3):
        for kx in range(3):
            patch = xp[:, :, ky:ky + stride * (ho - 1) + 1:stride, kx:kx + stride * (wo - 1) + 1:stride]
            e += np.einsum("bchw,oc->bohw", patch, wt[:, :, ky, kx].astype(np.float64))
    e += b.astype(np.float64)[None, :, None, None]
    if leaky:
        e = np.maximum(e, 0.2 * e)
    errs = [np.abs(res[o] - e).max() / scale for o in (0, 1)]
    ok = np.isfinite(res[1]).all() and errs[1] < 4 * max(errs[0], 2e-6)
    bad += not ok
    print("B%d %d->%d %dx%d s%d leaky=%d scale %g | fp32 kernel max err %.2e | bf16x6 %.2e %s" % (B, ci, co, h, w, stride, leaky, scale, errs[0], errs[1], "" if ok else "  <-- BAD"))
print("bad cases", bad)
assert bad == 0

# ---- the fused head (b2f_op_conv_head16) against fp64 and against the two fp32 kernels chained ----
bad = 0
for (B, h, w, scale) in [(1, 32, 60, 1.0), (2, 37, 71, 1.0), (1, 64, 122, 20.0), (3, 5, 9, 1.0), (1, 130, 61, 1e-2), (1, 256, 64, 1.0)]:
    x = (rng.standard_normal((B, 16, h, w)) * scale).astype(np.float32)
    w1 = (rng.standard_normal((16, 16, 3, 3)) / 12).astype(np.float32)
    w2 = (rng.standard_normal((32, 16, 3, 3)) / 12).astype(np.float32)
    b1 = (rng.standard_normal(16) * scale).astype(np.float32)
    b2 = (rng.standard_normal(32) * scale).astype(np.float32)
    y = ops.conv_head16(m, x, w1, b1, w2, b2)
    m.set_option("bf16_direct", 0)
    y32 = ops.conv3x3(m, ops.conv3x3(m, x, w1, b1, 1, True), w2, b2, 2, True)
    def conv64(xx, ww, bb, stride):
        Bc, ci, hh, wwd = xx.shape
        co = ww.shape[0]
        ho, wo = (hh - 1) // stride + 1, (wwd - 1) // stride + 1
        xp = np.pad(xx, ((0, 0), (0, 0), (1, 1), (1, 1)))
        e = np.zeros((Bc, co, ho, wo))
        for ky in range(3):
            for kx in range(3):
                e += np.einsum("bchw,oc->bohw", xp[:, :, ky:ky + stride * (ho - 1) + 1:stride, kx:kx + stride * (wo - 1) + 1:stride], ww[:, :, ky, kx].astype(np.float64))
        e += bb.astype(np.float64)[None, :, None, None]
        return np.maximum(e, 0.2 * e)
    e = conv64(conv64(x.astype(np.float64), w1, b1, 1), w2, b2, 2)
    e1, e0 = np.abs(y - e).max() / scale, np.abs(y32 - e).max() / scale
    ok = np.isfinite(y).all() and y.shape == e.shape and e1 < 4 * max(e0, 2e-6)
    bad += not ok
    print("head B%d %dx%d scale %g | fp32 kernels chained max err %.2e | fused bf16x6 %.2e %s" % (B, h, w, scale, e0, e1, "" if ok else "  <-- BAD"))
    if not ok:
        d = np.abs(y - e) / scale > 1e-4
        idx = np.argwhere(d)
        print("   wrong: rows", sorted(set(idx[:, 2].tolist()))[:30], "cols", sorted(set(idx[:, 3].tolist()))[:40], "co", sorted(set(idx[:, 1].tolist()))[:33])
print("bad cases", bad)
assert bad == 0
