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


def _load():
    name, H, W = open(os.path.join(HERE, "meta.txt")).read().split()
    H, W = int(H), int(W)
    flow = np.fromfile(os.path.join(HERE, "flow.f64"), "<f8").reshape(2, H, W)
    fo = np.fromfile(os.path.join(HERE, "fwd_occ.u8"), np.uint8).reshape(1, H, W)
    bo = np.fromfile(os.path.join(HERE, "bwd_occ.u8"), np.uint8).reshape(1, H, W)
    from back2future_amd import flow_io
    sd = os.path.join(os.path.dirname(HERE), "samples")
    ims = [flow_io.load_image(os.path.join(sd, "frame_%04d.png" % i)) for i in (9, 10, 11)]
    assert ims[0].shape == (3, H, W)
    return name, ims, flow, fo, bo


def _weights():
    import ctypes as C
    from back2future_amd import _lib
    n, pf = C.c_longlong(), C.c_int()
    path = os.path.join(HERE, "weights.t7").encode()
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
