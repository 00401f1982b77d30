"""GPU: graph shapes of createModelMulti other than the shipped one (models/pwc.lua:88-121) through
b2f_init_ex -- the generic executor of libb2f.so against the oracle's orc_pwc_forward_ex on the same weights."""
import numpy as np
import pytest

from back2future_amd import back2future, weights as W
from oracle import oracle as O
from tests.test_graph_options_cpu import CASES, _oracle_opts

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("past", [False, True])
def test_graph_options_full_table_vs_oracle(name, past):
    o = W.graph_opts(**CASES[name])
    m = back2future.Model("random:%s:11:2.0" % ("soft" if past else "hard"), graph=W.opts_string(o))
    try:
        assert (m.levels, m.win, m.past_flow) == (o["levels"], o["win"], past)
        assert m.n_outputs == (o["levels"] - o["skip"]) * (5 if past else 4)
        p = W.random_init(11, past, 2.0, o)
        np.testing.assert_array_equal(m.get_weights(), p)         # same generator, same canonical order of this graph
        # a bias on the last layer of every flow decoder: every level predicts a flow of several pixels (in units of its own map), so
        # the feature / image warps sample whole pixels away and through the border clamp -- random weights alone move by < 1 px
        for lname, shape, off in W.layout(past, o)[0]:
            if lname.endswith(".conv6.b") and (".flow." in lname or ".past." in lname):
                p[off:off + 2] = np.asarray((0.35, -0.25), np.float32) * (1.0 if ".flow." in lname else -1.0)
        m.set_weights(p)
        rng = np.random.default_rng(len(name) + 7 * past)
        mlt = 1 << (o["levels"] - 1)
        H, Wd = 4 * mlt, 6 * mlt
        x = rng.standard_normal((2, 9, H, Wd)).astype(np.float32)
        got = m.forward(x)
        exp = O.pwc_forward(x, p, past, _oracle_opts(o, past))
        assert len(got) == len(exp)
        assert np.abs(exp[0]).max() > 0.3
        for i, (a, b) in enumerate(zip(got, exp)):
            assert a.shape == b.shape, (i, a.shape, b.shape)
            assert np.isfinite(a).all()
            if a.shape[1] == 3:
                # warped images may differ by more at sampling-cell boundaries of a 1e-6 flow difference
                assert np.abs(a - b).max() <= 1e-3, (name, i, float(np.abs(a - b).max()))
            else:
                # flows / occlusion probabilities: fp32 re-association of the Winograd transforms only (a wrong channel offset in a
                # zero-padded JoinTable or a wrong flow-scale constant at some level would be orders of magnitude above this)
                np.testing.assert_allclose(a, b, rtol=2e-4, atol=3e-5, err_msg="%s output %d" % (name, i))
    finally:
        m.close()


def test_create_model_multi_nil_through_compute_flow():
    """createModelMulti(nil) (win 5, levels 4, pwc.lua:88) behind the computeFlow boundary: est[1] of the generic
    executor, post-processed as back2future.lua:77-93."""
    o = W.graph_opts(win=5, levels=4)
    m = back2future.Model("random:soft:3:2.0", graph=W.opts_string(o))
    try:
        rng = np.random.default_rng(5)
        ims = [rng.random((3, 128, 192), dtype=np.float32) for _ in range(3)]
        flow, fo, bo = m.computeFlow(*ims)
        xn = back2future.normalize(np.concatenate(ims, 0))[None]
        table = O.pwc_forward(xn, W.random_init(3, True, 2.0, o), True, _oracle_opts(o, True))
        assert np.abs(table[0][0]).max() > 0.02
        assert np.abs(flow - table[0][0].astype(np.float64)).max() <= 1e-3
        near = np.abs(table[2][0] - 0.6666) < 1e-3
        assert ((fo[0] != (table[2][0][1] >= 0.6666)) & ~near[1]).sum() == 0
        assert ((bo[0] != (table[2][0][0] >= 0.6666)) & ~near[0]).sum() == 0
    finally:
        m.close()


def test_shipped_graph_is_unaffected_by_an_empty_option_string():
    a = back2future.Model("random:hard:5:2.0", graph="")
    b = back2future.Model("random:hard:5:2.0", graph="win=9,levels=7,skip=2")
    try:
        rng = np.random.default_rng(1)
        ims = [rng.random((3, 64, 128), dtype=np.float32) for _ in range(3)]
        for u, v in zip(a.computeFlow(*ims), b.computeFlow(*ims)):
            np.testing.assert_array_equal(u, v)
    finally:
        a.close()
        b.close()


def test_non_shipped_graph_from_a_t7_file(tmp_path):
    """b2f_init on a .t7 that holds createModelMulti(nil) (win 5, levels 4): the graph shape comes out of the file (no option
    string), the context computes what the same weights give through the option-string route, bit for bit."""
    from tests import t7_writer
    o = W.graph_opts(win=5, levels=4)
    flat = W.random_init(3, True, 2.0, o)
    p = str(tmp_path / "nil.t7")
    with open(p, "wb") as f:
        t7_writer.Writer(f).obj(t7_writer.build_model(flat, True, o=o))
    a = back2future.Model(p)
    b = back2future.Model("random:soft:3:2.0", graph=W.opts_string(o))
    try:
        assert (a.levels, a.win, a.past_flow) == (4, 5, True)
        np.testing.assert_array_equal(a.get_weights(), flat)
        rng = np.random.default_rng(9)
        ims = [rng.random((3, 128, 192), dtype=np.float32) for _ in range(3)]
        for u, v in zip(a.computeFlow(*ims), b.computeFlow(*ims)):
            np.testing.assert_array_equal(u, v)
        with pytest.raises(Exception, match="window"):
            back2future.Model(p, graph="win=9,levels=4")
    finally:
        a.close()
        b.close()


def test_skip0_graph_from_a_t7_file_through_compute_flow(tmp_path):
    """pwc_skip = 0 (pwc.lua:120-122,171-173,423-429,462-471) end to end: the graph shape out of a .t7 (level-1 convUnit, the equally
    wide decoders of levels 1 and 2 told apart by the node order), est[1] = fs[1] / est[3] = occs[1] at full resolution through the
    computeFlow boundary against the oracle's output table."""
    from tests import t7_writer
    o = W.graph_opts(win=3, levels=4, skip=0)
    flat = W.random_init(5, True, 2.0, o)
    p = str(tmp_path / "skip0.t7")
    with open(p, "wb") as f:
        t7_writer.Writer(f).obj(t7_writer.build_model(flat, True, o=o))
    m = back2future.Model(p)
    try:
        assert (m.levels, m.win, m.past_flow, m.n_outputs) == (4, 3, True, 20)
        np.testing.assert_array_equal(m.get_weights(), flat)
        rng = np.random.default_rng(2)
        ims = [rng.random((3, 64, 128), dtype=np.float32) for _ in range(3)]
        flow, fo, bo = m.computeFlow(*ims)
        xn = back2future.normalize(np.concatenate(ims, 0))[None]
        table = O.pwc_forward(xn, flat, True, _oracle_opts(o, True))
        assert table[0].shape == (1, 2, 64, 128) and np.abs(table[0][0]).max() > 0.02
        assert np.abs(flow - table[0][0].astype(np.float64)).max() <= 1e-3
        near = np.abs(table[2][0] - 0.6666) < 1e-3
        assert ((fo[0] != (table[2][0][1] >= 0.6666)) & ~near[1]).sum() == 0
        assert ((bo[0] != (table[2][0][0] >= 0.6666)) & ~near[0]).sum() == 0
    finally:
        m.close()
