"""The one true end-to-end pin this repository can take: outputs of the reference itself,
dumped out of band by tests/golden/torch7_dump/dump_samples.lua on a Torch7 machine with the pretrained weights.
Both tests skip until tests/golden/torch7_dump/ holds flow.f64, fwd_occ.u8, bwd_occ.u8, meta.txt and weights.t7;
dropping the files in is all it takes to turn 'parity unpinned' into a pinned comparison."""
import os

import numpy as np
import pytest

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "torch7_dump")
NEEDED = ["flow.f64", "fwd_occ.u8", "bwd_occ.u8", "meta.txt", "weights.t7"]
have = all(os.path.exists(os.path.join(HERE, f)) for f in NEEDED)
reason = "no Torch7 dump under tests/golden/torch7_dump (see dump_samples.lua there)"


def _load(here=HERE, samples=None):
    name, H, W = open(os.path.join(here, "meta.txt")).read().split()
    H, W = int(H), int(W)
    flow = np.fromfile(os.path.join(here, "flow.f64"), "<f8").reshape(2, H, W)
    fo = np.fromfile(os.path.join(here, "fwd_occ.u8"), np.uint8).reshape(1, H, W)
    bo = np.fromfile(os.path.join(here, "bwd_occ.u8"), np.uint8).reshape(1, H, W)
    from back2future_amd import flow_io
    sd = samples or os.path.join(os.path.dirname(HERE), "samples")
    ims = [flow_io.load_image(os.path.join(sd, "frame_%04d.png" % i)) for i in (9, 10, 11)]
    assert ims[0].shape == (3, H, W)
    return name, ims, flow, fo, bo


def _weights(here=HERE):
    import ctypes as C
    from back2future_amd import _lib
    n, pf = C.c_longlong(), C.c_int()
    path = os.path.join(here, "weights.t7").encode()
    _lib.check(_lib.lib().b2f_load_t7(path, None, 0, C.byref(n), C.byref(pf)))
    w = np.empty(n.value, np.float32)
    _lib.check(_lib.lib().b2f_load_t7(path, _lib.fptr(w), w.size, C.byref(n), C.byref(pf)))
    return w, bool(pf.value)


def _compare(flow, fo, bo, rflow, rfo, rbo):
    d = np.abs(flow - rflow)
    epe = np.sqrt(((flow - rflow) ** 2).sum(0)).mean()
    assert d.max() <= 1e-3 and epe <= 1e-3, (float(d.max()), float(epe))
    # masks: thresholds of est[3]; a pixel whose probability sits within 1e-3 of 0.6666 may flip -- allow 0.1 % of pixels
    assert (fo != rfo).mean() <= 1e-3 and (bo != rbo).mean() <= 1e-3


@pytest.mark.skipif(not have, reason=reason)
def test_oracle_against_the_torch7_dump():
    from oracle import oracle as O
    name, ims, rflow, rfo, rbo = _load()
    w, past = _weights()
    flow, fo, bo = O.compute_flow(*ims, w, past)
    _compare(flow, fo, bo, rflow, rfo, rbo)


@pytest.mark.gpu
@pytest.mark.skipif(not have, reason=reason)
def test_library_against_the_torch7_dump():
    from back2future_amd import back2future
    name, ims, rflow, rfo, rbo = _load()
    computeFlow = back2future.init(os.path.join(HERE, "weights.t7"))
    flow, fo, bo = computeFlow(*ims)
    _compare(flow, fo, bo, rflow, rfo, rbo)


def test_the_dump_loaders_on_a_synthetic_dump(tmp_path):
    """Keeps the file-drop pin alive until a real dump arrives: a dump in dump_samples.lua's format (raw little-endian flow.f64,
    fwd_occ.u8, bwd_occ.u8, meta.txt) is written here from the ORACLE's own output on a crop of the sample frames, with the weights
    as a Torch7-serialized weights.t7 (tests/t7_writer.py), and goes through exactly the loaders and the comparison the real
    dump will go through -- a slip in _load() / _weights() / _compare() (shapes, dtypes, byte order, mask orientation, the
    thresholds) shows now, not on the day the files are dropped in.  It pins nothing about the reference."""
    from back2future_amd import flow_io, weights as Wt
    from oracle import oracle as O
    from tests import t7_writer
    sd = os.path.join(os.path.dirname(HERE), "samples")
    H, W = 100, 170                                  # not a multiple of 64: the image.scale legs of computeFlow run too
    frames = [flow_io.load_image(os.path.join(sd, "frame_%04d.png" % i))[:, 40:40 + H, 300:300 + W] for i in (9, 10, 11)]
    samples = tmp_path / "samples"
    dump = tmp_path / "dump"
    samples.mkdir(); dump.mkdir()
    for i, f in zip((9, 10, 11), frames):
        flow_io.save_image(str(samples / ("frame_%04d.png" % i)), f)
    w = Wt.random_init(4, True, 2.0)
    for lname, shape, off in Wt.layout(True)[0]:
        if lname == "l3.occ.conv6.b":
            w[off:off + 2] = (0.7, 0.0)              # occlusion logits around the 0.6666 threshold of channel 0: mixed masks
    t7_writer.save(str(dump / "weights.t7"), w, True)
    ims = [flow_io.load_image(str(samples / ("frame_%04d.png" % i))) for i in (9, 10, 11)]   # 8-bit round trip, as image.load sees it
    flow, fo, bo = O.compute_flow(*ims, w, True)
    assert flow.dtype == np.float64 and flow.shape == (2, H, W) and fo.shape == (1, H, W) and fo.dtype == np.uint8
    flow.astype("<f8").tofile(str(dump / "flow.f64"))
    fo.tofile(str(dump / "fwd_occ.u8"))
    bo.tofile(str(dump / "bwd_occ.u8"))
    (dump / "meta.txt").write_text("Ours-Soft-ft-KITTI %d %d\n" % (H, W))
    name, ims2, rflow, rfo, rbo = _load(str(dump), str(samples))
    w2, past = _weights(str(dump))
    assert name == "Ours-Soft-ft-KITTI" and past is True
    np.testing.assert_array_equal(w2, w)             # the .t7 reader returns the canonical flat order
    for a, b in zip(ims, ims2):
        np.testing.assert_array_equal(a, b)
    _compare(*O.compute_flow(*ims2, w2, past), rflow, rfo, rbo)
    # and the comparison does reject a dump that is off: swapped masks, a flow that is 2e-3 away
    assert (rfo != rbo).mean() > 1e-3, "degenerate masks: the swap check below would be vacuous"
    with pytest.raises(AssertionError):
        _compare(flow, fo, bo, rflow, rbo, rfo)
    with pytest.raises(AssertionError):
        _compare(flow, fo, bo, rflow + 2e-3, rfo, rbo)


# ---------------------------------------------------------------------------------------------------------------------
# Op-level pins (tests/golden/torch7_dump/dump_ops.lua -> tests/golden/torch7_dump/ops/): every third-party op the oracle
# restates + a Torch-written model file with its whole output table.  Inputs are regenerated here with the script's
# counter generator; the same comparison code runs today on a SYNTHETIC dump written from the oracle (loaders, meta
# parsing, generator, .t7 route), and on the real one the day it is dropped in.
OPS = os.path.join(HERE, "ops")
have_ops = os.path.exists(os.path.join(OPS, "ops_meta.txt"))
reason_ops = "no Torch7 op dump under tests/golden/torch7_dump/ops (see dump_ops.lua there)"


def lcg(seed, *shape):
    """dump_ops.lua's generator: s <- (s * 69069 + 1) mod 2^32, value = s / 2^32 - 0.5 (float32), C order."""
    n = int(np.prod(shape))
    out = np.empty(n, np.float64)
    s = int(seed)
    for i in range(n):
        s = (s * 69069 + 1) % 4294967296
        out[i] = s / 4294967296.0 - 0.5
    return out.astype(np.float32).reshape(shape)


def _read_ops(d):
    out = {}
    for line in open(os.path.join(d, "ops_meta.txt")):
        f = line.split()
        if f:
            shape = tuple(int(v) for v in f[2:2 + int(f[1])])
            out[f[0]] = np.fromfile(os.path.join(d, f[0] + ".f32"), "<f4").reshape(shape)
    return out


def _op_inputs(samples=None):
    from back2future_amd import flow_io
    sd = samples or os.path.join(os.path.dirname(HERE), "samples")
    return {"ref": lcg(1001, 2, 16, 12, 20) * np.float32(2), "frm": lcg(1002, 2, 16, 12, 20) * np.float32(2),
            "img": lcg(3001, 2, 8, 12, 20), "flow": lcg(3002, 2, 2, 12, 20) * np.float32(12),
            "x": lcg(4001, 2, 2, 6, 10), "logits": lcg(5001, 2, 2, 6, 10) * np.float32(8),
            "frame": flow_io.load_image(os.path.join(sd, "frame_0009.png")), "model_in": lcg(8001, 1, 9, 64, 128)}


def _oracle_ops(inp):
    from oracle import oracle as O
    down = O.image_scale_bilinear(inp["frame"], 320, 1216)
    return {"costvol_fwd": O.costvol([inp["ref"], inp["frm"]], 9, True), "costvol_bwd": O.costvol([inp["ref"], inp["frm"]], 9, False),
            "warp": O.warping_unit(inp["img"], inp["flow"], 1.0),
            "upsample_bilinear2": O.upsample_bilinear2x(inp["x"]), "upsample_nearest2": O.upsample_nearest2x(inp["x"]),
            "softmax": O.spatial_softmax(inp["logits"]),
            "scale_bilinear": down,
            # the Lua script scales a DoubleTensor there, as computeFlow does with the flow (back2future.lua:77-80)
            "scale_simple": O.image_scale_simple(down[:2].astype(np.float64), inp["frame"].shape[1], inp["frame"].shape[2])}


def _load_model_file(path):
    """flat weights (canonical order), past_flow, option dict -- read from the file alone (graph shape inferred)."""
    import ctypes as C
    from back2future_amd import _lib, weights as Wt
    L = _lib.lib()
    n, buf = C.c_longlong(), C.create_string_buffer(512)
    _lib.check(L.b2f_load_t7_ex(path.encode(), None, None, 0, C.byref(n), buf, 512))
    w = np.empty(n.value, np.float32)
    _lib.check(L.b2f_load_t7_ex(path.encode(), None, _lib.fptr(w), w.size, C.byref(n), buf, 512))
    o = dict(Wt.SHIPPED)
    for kv in buf.value.decode().split(","):
        k, v = kv.split("=")
        if k in o:
            o[k] = type(o[k])(float(v))
    past = w.size == Wt.param_count(True, o)
    assert w.size == Wt.param_count(past, o)
    return w, past, o


def _oracle_model(d, x):
    from oracle import oracle as O
    w, past, o = _load_model_file(os.path.join(d, "tiny_model.t7"))
    oo = O.opts(past, win=o["win"], levels=o["levels"], skip=o["skip"], two_frame=o["two_frame"], sum_cvs=o["sum_cvs"],
                residual=o["residual"], occ_input=o["occ_input"], rescale_flow=o["rescale_flow"], flownet_factor=o["flownet_factor"])
    return O.pwc_forward(x, w, past, oo), o


def _compare_ops(d, samples=None):
    """Oracle against an op dump in directory d; returns the number of tensors compared."""
    ref = _read_ops(d)
    inp = _op_inputs(samples)
    got = _oracle_ops(inp)
    n = 0
    for name, g in got.items():
        assert name in ref, name
        assert ref[name].shape == g.shape, (name, ref[name].shape, g.shape)
        # fp32 ops with a handful of roundings each; bit-exactness is not claimed across implementations
        np.testing.assert_allclose(g, ref[name], rtol=1e-5, atol=1e-5, err_msg=name)
        n += 1
    if os.path.exists(os.path.join(d, "tiny_model.t7")):
        outs, o = _oracle_model(d, inp["model_in"])
        ests = sorted(k for k in ref if k.startswith("est_"))
        assert len(ests) == len(outs), (len(ests), len(outs))            # the est table: count, order and sizes
        for k, g in zip(ests, outs):
            assert ref[k].shape == g.shape, (k, ref[k].shape, g.shape)
            np.testing.assert_allclose(g, ref[k], rtol=0, atol=2e-4, err_msg=k)
            n += 1
    return n


@pytest.mark.skipif(not have_ops, reason=reason_ops)
def test_oracle_against_the_torch7_op_dump():
    assert _compare_ops(OPS) >= 8


@pytest.mark.gpu
@pytest.mark.skipif(not have_ops, reason=reason_ops)
def test_library_against_the_torch7_op_dump():
    from back2future_amd import back2future, ops
    ref = _read_ops(OPS)
    inp = _op_inputs()
    m = back2future.Model("random:hard:1:1.0")
    try:
        for fwd, name in ((True, "costvol_fwd"), (False, "costvol_bwd")):
            np.testing.assert_allclose(ops.costvol(m, inp["ref"], inp["frm"], 9, fwd), ref[name], rtol=1e-5, atol=1e-5, err_msg=name)
        grid = np.ascontiguousarray(inp["flow"].transpose(0, 2, 3, 1))
        img = np.ascontiguousarray(inp["img"].transpose(0, 2, 3, 1))
        np.testing.assert_allclose(ops.warp_bhwd(m, img, grid).transpose(0, 3, 1, 2), ref["warp"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(ops.upsample_flow2x(m, inp["x"]), ref["upsample_bilinear2"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(ops.image_scale(m, inp["frame"], 320, 1216), ref["scale_bilinear"], rtol=1e-6, atol=1e-6)
    finally:
        m.close()
    path = os.path.join(OPS, "tiny_model.t7")
    if os.path.exists(path):
        tm = back2future.Model(path)               # graph shape read from the Torch-written file
        try:
            outs = tm.forward(inp["model_in"])
            ests = sorted(k for k in ref if k.startswith("est_"))
            assert len(ests) == len(outs)
            for k, g in zip(ests, outs):
                np.testing.assert_allclose(g, ref[k], rtol=0, atol=2e-4, err_msg=k)
        finally:
            tm.close()


def test_the_op_dump_loaders_on_a_synthetic_dump(tmp_path):
    """A dump in dump_ops.lua's format written from the ORACLE's own outputs (and a createModelMulti(nil)-shaped model file
    written by tests/t7_writer.py) goes through the loaders, the input generator and the comparison the real dump will go
    through; a corrupted tensor and a truncated est table are rejected.  It pins nothing about the reference."""
    from back2future_amd import weights as Wt
    from oracle import oracle as O
    from tests import t7_writer
    d = tmp_path / "ops"
    d.mkdir()
    inp = _op_inputs()
    assert abs(float(inp["ref"].mean())) < 0.05 and float(inp["ref"].std()) > 0.5        # the generator is not degenerate
    assert np.array_equal(lcg(1001, 5), lcg(1001, 2, 16, 12, 20).ravel()[:5])
    got = _oracle_ops(inp)
    o = Wt.graph_opts(win=5, levels=4)                                                   # createModelMulti(nil), pwc.lua:88-100
    w = Wt.random_init(7, False, 2.0, o)
    t7_writer.save(str(d / "tiny_model.t7"), w, False, o=o)
    oo = O.opts(False, win=5, levels=4)
    outs = O.pwc_forward(inp["model_in"], w, False, oo)
    for i, e in enumerate(outs):
        got["est_%02d" % (i + 1)] = e
    with open(str(d / "ops_meta.txt"), "w") as meta:
        for name, t in got.items():
            np.ascontiguousarray(t, "<f4").tofile(str(d / (name + ".f32")))
            meta.write("%s %d %s\n" % (name, t.ndim, " ".join(str(s) for s in t.shape)))
    w2, past, o2 = _load_model_file(str(d / "tiny_model.t7"))
    assert past is False and o2["win"] == 5 and o2["levels"] == 4 and o2["skip"] == 2
    np.testing.assert_array_equal(w2, w)
    assert _compare_ops(str(d)) == 8 + len(outs)
    # a dump that is off must be rejected: one corrupted op, one missing est entry
    bad = got["warp"].copy(); bad[0, 0, 0, 0] += 1e-3
    np.ascontiguousarray(bad, "<f4").tofile(str(d / "warp.f32"))
    with pytest.raises(AssertionError):
        _compare_ops(str(d))
    np.ascontiguousarray(got["warp"], "<f4").tofile(str(d / "warp.f32"))
    lines = [l for l in open(str(d / "ops_meta.txt")) if not l.startswith("est_%02d" % len(outs))]
    open(str(d / "ops_meta.txt"), "w").writelines(lines)
    with pytest.raises(AssertionError):
        _compare_ops(str(d))
