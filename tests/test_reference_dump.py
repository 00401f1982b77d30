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
