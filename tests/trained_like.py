"""Weights that BEHAVE like trained ones, since the real ones cannot be had (the .t7 files of back2future.lua:100-113 are Dropbox links).

A random-init model's activations shrink layer by layer (Torch's reset() scale with LeakyReLU(0.2) loses a factor ~6 of variance per conv),
so the decoders of a random model work on tiny numbers and every kernel's rounding looks harmless.  `calibrate` rescales every convolution
of the pruned computeFlow graph (weights and bias) so that its output has unit standard deviation on a given input -- what batch statistics
of a trained network look like.  `add_outliers`, applied BEFORE it, multiplies a few output filters per layer by 10: after the calibration
every layer still has unit deviation overall, carried by 3 % of its channels at 10 x the rest (heavy-tailed channels, the case Winograd
transforms like least).  Test infrastructure: the PyTorch-CPU statement of the graph (oracle/torch_cpu.py) does the walking."""
import numpy as np


def calibrate(params, x, past_flow, target=1.0, target_flow=0.25):
    """params: canonical flat weights; x: B x 9 x H x W normalized input.  Returns a rescaled copy.  The 2-output heads (flow / 20, occlusion
    logits) get deviation `target_flow`: flows of a few units = tens of pixels at full resolution, as in KITTI; at deviation 1 the warps
    amplify ANY fp32 difference (the oracle's own included) to the contract's 1e-3 and the comparison measures that, not the kernels."""
    import torch
    import torch.nn.functional as F
    from oracle import torch_cpu as T
    p = np.ascontiguousarray(np.array(params, dtype=np.float32, copy=True))
    seen = set()
    orig = T._conv

    def conv(xx, wb, stride=1, leaky=True):
        key = wb[0].data_ptr()
        if key not in seen:                       # (the siamese towers apply the same layer three times: the first frame sets its scale)
            seen.add(key)
            s = float(F.conv2d(xx, wb[0], wb[1], stride=stride, padding=1).std())
            if s > 0:
                t = target_flow if wb[0].shape[0] == 2 else target
                wb[0].mul_(t / s)
                wb[1].mul_(t / s)
        return orig(xx, wb, stride, leaky)
    T._conv = conv
    try:
        with torch.no_grad():
            T.compute_flow_graph(x, p, past_flow)   # _split() views the array p itself (contiguous float32: no copy)
    finally:
        T._conv = orig
    assert len(seen) >= 12 + 5 * 6 + 6
    return p


def add_outliers(params, past_flow, frac=0.03, factor=10.0, seed=0):
    """Multiplies `frac` of the output filters (weights and bias) of every layer with >= 32 outputs by `factor`."""
    from back2future_amd import weights as W
    p = np.array(params, dtype=np.float32, copy=True)
    r = np.random.default_rng(seed)
    lay = W.layout(past_flow)[0]
    by_name = {n: (shape, off) for n, shape, off in lay}
    for name, shape, off in lay:
        if not name.endswith(".w") or shape[0] < 32:
            continue
        co = shape[0]
        hot = r.random(co) < frac
        if not hot.any():
            continue
        per = int(np.prod(shape[1:]))
        w = p[off:off + co * per].reshape(co, per)
        w[hot] *= factor
        bshape, boff = by_name[name[:-2] + ".b"]
        p[boff:boff + co][hot] *= factor
    return p
