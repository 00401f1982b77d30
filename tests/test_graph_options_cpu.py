"""CPU: the option table of createModelMulti (models/pwc.lua:88-121) -- SURVEY s8 f4 -- in the oracle
(orc_pwc_forward_ex) against the PyTorch witness (tests/torch_ref.py) branch by branch, and the host-side layout /
option parsing of libb2f.so against both."""
import ctypes as C

import numpy as np
import pytest
import torch

from back2future_amd import _lib, weights as W
from oracle import oracle as O
from tests import torch_ref as R

torch.set_num_threads(4)

CASES = {
    "createModelMulti(nil)": dict(win=5, levels=4),                                   # pwc.lua:88
    "sum_cvs": dict(win=5, levels=5, sum_cvs=1),                                      # :266-276
    "residual": dict(win=3, levels=5, residual=1),                                    # :341-351
    "occ_input+rescale": dict(win=5, levels=5, occ_input=1, rescale_flow=1),          # :300-304, :364-369
    "two_frame": dict(win=5, levels=4, two_frame=1),                                  # :160-165, :279-284, :292-296
    "skip1+factor": dict(win=3, levels=4, skip=1, flownet_factor=10.0),               # :136, :404
    "skip3": dict(win=3, levels=5, skip=3, residual=1, sum_cvs=1),
    "skip0": dict(win=3, levels=3, skip=0),                                           # :120-122,171-173,359,423-429,462-471
    "skip0+occ_input+rescale": dict(win=3, levels=3, skip=0, occ_input=1, rescale_flow=1),
    "no_siamese": dict(win=5, levels=4, siamese=0),                                   # :125-127,175,182
    "skip0+no_siamese+two_frame": dict(win=3, levels=3, skip=0, siamese=0, two_frame=1),
}


def _oracle_opts(o, past):
    return O.opts(past, win=o["win"], levels=o["levels"], skip=o["skip"], two_frame=o["two_frame"], sum_cvs=o["sum_cvs"],
                  residual=o["residual"], occ_input=o["occ_input"], rescale_flow=o["rescale_flow"],
                  flownet_factor=o["flownet_factor"], siamese=o["siamese"])


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("past", [False, True])
def test_oracle_graph_options_vs_torch(name, past):
    o = W.graph_opts(**CASES[name])
    oo = _oracle_opts(o, past)
    assert O.param_count(past, oo) == W.param_count(past, o)
    rng = np.random.default_rng(len(name) + past)
    m = 1 << (o["levels"] - 1)
    H, Wd = 2 * m, 3 * m
    x = rng.standard_normal((2, 9, H, Wd)).astype(np.float32)
    p = W.random_init(11, past, 2.0, o)
    got = O.pwc_forward(x, p, past, oo)
    exp, _ = R.pwc_forward(x, W.views(p, past, o), past, o=o)
    assert len(got) == len(exp) == (o["levels"] - o["skip"]) * (5 if past else 4)
    assert max(np.abs(g).max() for g in got[:1]) > 0.02
    for i, (a, b) in enumerate(zip(got, exp)):
        assert a.shape == b.shape, (i, a.shape, b.shape)
        np.testing.assert_allclose(a, b, rtol=0, atol=2e-4, err_msg="%s output %d" % (name, i))


def test_shipped_options_are_the_default_graph():
    for past in (False, True):
        assert O.param_count(past, O.opts(past)) == O.param_count(past) == W.param_count(past, W.graph_opts())


def test_pruned_equals_full_where_compute_flow_reads():
    rng = np.random.default_rng(3)
    x = rng.standard_normal((1, 9, 64, 128)).astype(np.float32)
    for past in (False, True):
        p = W.random_init(4, past, 2.0)
        full = O.pwc_forward(x, p, past)
        pr = O.pwc_forward(x, p, past, pruned=True)
        np.testing.assert_array_equal(pr[0], full[0])                       # est[1]
        np.testing.assert_array_equal(pr[2 if past else 1], full[2 if past else 1])   # skip_occs[3]
        if not past:
            np.testing.assert_array_equal(pr[2], full[2])                   # Hard: est[3] = iws[1][3]


def test_library_host_side_agrees_on_layouts():
    """b2f_init_ex's option parser / layout are host code: exercised through error paths that need no GPU ... the
    parameter counts are checked on the GPU box (tests/test_gpu_graph_options.py)."""
    L = _lib.lib()
    h = C.c_void_p()
    rc = L.b2f_init_ex(b"random:hard", 0, b"win=4", C.byref(h))
    assert rc != 0
    msg = L.b2f_last_error().decode()
    assert "unsupported graph options" in msg or "no HIP device" in msg
    rc = L.b2f_init_ex(b"random:hard", 0, b"bogus=1", C.byref(h))
    assert rc != 0 and ("unknown graph option" in L.b2f_last_error().decode())
