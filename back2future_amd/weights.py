"""Canonical flat weight layout of the shipped multi-frame PWC graph and a
deterministic random initialiser.

Layout (fp32, little endian), mirroring how models/pwc.lua builds the graph:
  feature units l = 2..7 (convUnit, pwc.lua:58-65):  conv1.w, conv1.b, conv2.w, conv2.b
  then for l = 7..3:  occlusion decoder, future-flow decoder, [past-flow decoder]
                      (decoder(), pwc.lua:76-85), each 6 x {w, b}
every w in Torch layout Co x Ci x 3 x 3.  Decoder input-channel order is the
JoinTable order of pwc.lua:308,334,337: {cost volume (fwd 81, bwd 81), cs[ref][l],
upsampled flow}.  7 193 316 floats ("Ours-Hard", past_flow=false) or
10 168 302 ("Ours-Soft-*", past_flow=true).

The same splitmix64 counter generator is implemented in csrc/b2f_weights.cpp
(b2f_random_weights); tests/test_weights.py checks they agree bit for bit.
"""
import numpy as np

FEAT = [0, 3, 16, 32, 64, 96, 128, 192]   # featMaps, pwc.lua:29,89 (index = level)
DEC = [128, 128, 96, 64, 32, 2]           # decoder(), pwc.lua:76-85
LEVELS, L_ST, WIN = 7, 3, 9
ND = 2 * WIN * WIN


SHIPPED = dict(win=9, levels=7, skip=2, two_frame=0, sum_cvs=0, residual=0, occ_input=0, rescale_flow=0, flownet_factor=20.0, siamese=1)


def graph_opts(**kw):
    """createModelMulti option table (pwc.lua:88-121): the shipped values (opts.lua:83-98) with overrides."""
    o = dict(SHIPPED)
    for k, v in kw.items():
        assert k in o, k
        o[k] = v
    return o


def opts_string(o):
    """The graph_opts argument of b2f_init_ex for an option dict."""
    return ",".join("%s=%g" % (k, float(v)) for k, v in o.items())


def feat_ch(l, o=SHIPPED):
    """featMaps[l] of createModelMulti (pwc.lua:89,120-127): pwc_skip = 0 gives the level-1 unit 16 maps, pwc_siamese = 0 replaces
    the learned pyramid by the (average-pooled) image: 3 maps on every level."""
    if not o.get("siamese", 1):
        return 3
    if l == 1 and o["skip"] == 0:
        return FEAT[2]
    return FEAT[l]


def occ_in_ch(l, o=SHIPPED):
    nd = o["win"] ** 2
    n = (nd if o["two_frame"] else 2 * nd) + feat_ch(l, o) + (feat_ch(l, o) if o["two_frame"] else 0)
    if l != o["levels"]:
        n += 2 + (2 if o["occ_input"] else 0)
    return n


def flow_in_ch(l, o=SHIPPED):
    nd = o["win"] ** 2
    ndf = nd if (o["two_frame"] or o["sum_cvs"]) else 2 * nd
    return ndf if l == o["levels"] else ndf + feat_ch(l, o) + 2


def layout(past_flow, o=SHIPPED):
    """List of (name, shape, offset) in canonical order."""
    out, off = [], 0
    LEVELS, L_ST = o["levels"], max(o["skip"] + 1, 1)   # pwc.lua:136

    def add(name, shape):
        nonlocal off
        out.append((name, tuple(shape), off))
        off += int(np.prod(shape))

    if o.get("siamese", 1):                               # pwc.lua:169-183: the level-1 unit only with pwc_skip = 0, none without the siamese net
        for l in range(1 if o["skip"] == 0 else 2, LEVELS + 1):
            ci, co = (3 if l == 1 else feat_ch(l - 1, o)), feat_ch(l, o)
            add("feat%d.conv1.w" % l, (co, ci, 3, 3)); add("feat%d.conv1.b" % l, (co,))
            add("feat%d.conv2.w" % l, (co, co, 3, 3)); add("feat%d.conv2.b" % l, (co,))
    for l in range(LEVELS, L_ST - 1, -1):
        kinds = [("occ", occ_in_ch(l, o)), ("flow", flow_in_ch(l, o))]
        if past_flow:
            kinds.append(("past", flow_in_ch(l, o)))
        for kind, n in kinds:
            ci = n
            for i, co in enumerate(DEC):
                add("l%d.%s.conv%d.w" % (l, kind, i + 1), (co, ci, 3, 3))
                add("l%d.%s.conv%d.b" % (l, kind, i + 1), (co,))
                ci = co
    return out, off


def param_count(past_flow, o=SHIPPED):
    return layout(past_flow, o)[1]


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def uniform01(seed, n, stream=0):
    """n floats in [0,1): 24 high bits of splitmix64(seed*2^32 + stream*2^40.. + i)."""
    with np.errstate(over="ignore"):
        base = np.uint64(seed) * np.uint64(0x100000001B3) + np.uint64(stream) * np.uint64(0xD6E8FEB86659FD93)
        idx = np.arange(n, dtype=np.uint64) + base
        z = _splitmix64(idx)
    return ((z >> np.uint64(40)).astype(np.float32)) * np.float32(1.0 / 16777216.0)


def random_init(seed=2, past_flow=False, gain=1.0, o=SHIPPED):
    """nn.SpatialConvolution:reset() [3P]: weight and bias ~ U(-s, s), s = 1/sqrt(9*Ci);
    `gain` scales s (tests use gain > 1 so that flows are O(1) and warps matter)."""
    lay, total = layout(past_flow, o)
    u = uniform01(seed, total)
    w = np.empty(total, np.float32)
    fan_in = None
    for name, shape, off in lay:
        n = int(np.prod(shape))
        if name.endswith(".w"):
            fan_in = shape[1] * 9
        s = np.float32(gain) / np.sqrt(np.float32(fan_in))
        w[off:off + n] = (np.float32(2.0) * u[off:off + n] - np.float32(1.0)) * s
    return w


def views(flat, past_flow, o=SHIPPED):
    lay, total = layout(past_flow, o)
    assert flat.size == total, (flat.size, total)
    return {name: flat[off:off + int(np.prod(shape))].reshape(shape) for name, shape, off in lay}
