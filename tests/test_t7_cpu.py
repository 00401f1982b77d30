"""CPU: the .t7 reader (b2f_load_t7, host-only) against files produced by tests/t7_writer.py."""
import ctypes as C
import os

import numpy as np
import pytest

from back2future_amd import _lib, build, weights as W
from tests import t7_writer


@pytest.fixture(scope="module", autouse=True)
def built():
    build.build()


def load_t7(path):
    L = _lib.lib()
    n, pf = C.c_longlong(), C.c_int()
    _lib.check(L.b2f_load_t7(path.encode(), None, 0, C.byref(n), C.byref(pf)))
    out = np.empty(n.value, np.float32)
    _lib.check(L.b2f_load_t7(path.encode(), _lib.fptr(out), out.size, C.byref(n), C.byref(pf)))
    return out, bool(pf.value)


@pytest.mark.parametrize("past_flow,cuda,cudnn,dpt", [(False, True, True, False), (True, True, True, True),
                                                      (True, False, False, False), (False, False, True, True)])
def test_roundtrip(tmp_path, past_flow, cuda, cudnn, dpt):
    flat = W.random_init(3, past_flow, 1.0)
    p = str(tmp_path / "model.t7")
    t7_writer.save(p, flat, past_flow, cuda=cuda, cudnn=cudnn, dpt=dpt)
    got, pf = load_t7(p)
    assert pf == past_flow
    np.testing.assert_array_equal(got, flat)          # every conv landed in its canonical slot, bit exact
    # file size ~ 2x parameters (weights + grads), as a checkpoint after clearState() (train.lua:181)
    assert os.path.getsize(p) > 2 * 4 * flat.size


def test_decoder_roles_are_not_assigned_by_shape(tmp_path):
    """occ / flow / past decoders of levels 3..6 have identical shapes: give each distinct weights
    and check they are told apart through the graph edges."""
    flat = W.random_init(9, True, 1.0)
    v = W.views(flat, True)
    v["l4.occ.conv1.w"][...] = 1.0; v["l4.flow.conv1.w"][...] = 2.0; v["l4.past.conv1.w"][...] = 3.0
    p = str(tmp_path / "m.t7")
    t7_writer.save(p, flat, True)
    got, _ = load_t7(p)
    g = W.views(got, True)
    assert g["l4.occ.conv1.w"].min() == 1.0 and g["l4.flow.conv1.w"].min() == 2.0 and g["l4.past.conv1.w"].min() == 3.0


def test_errors(tmp_path):
    with pytest.raises(_lib.B2FError, match="cannot open"):
        load_t7(str(tmp_path / "missing.t7"))
    bad = tmp_path / "bad.t7"
    bad.write_bytes(b"\x09\x00\x00\x00garbage")
    with pytest.raises(_lib.B2FError, match="not a readable"):
        load_t7(str(bad))
    trunc = tmp_path / "trunc.t7"
    t7_writer.save(str(trunc), W.random_init(3, False, 1.0), False)
    data = trunc.read_bytes()
    trunc.write_bytes(data[:len(data) // 3])
    with pytest.raises(_lib.B2FError):
        load_t7(str(trunc))


@pytest.mark.parametrize("past_flow,versioned,replicas", [(False, False, 2), (True, True, 3), (True, False, 1)])
def test_structurally_different_serialization(tmp_path, past_flow, versioned, replicas):
    """tests/t7_writer.py:build_model_legacy: unversioned object headers, nn.* on FloatTensors, every parameter a view
    (offset, 2-D weight, strided bias) into one flat storage, unknown fields, numeric table keys, serialized functions
    of tags 6 / 7 / 8 (tag 6 carries no reference index), reversed forward nodes, a DataParallelTable with several
    replicas."""
    flat = W.random_init(13, past_flow, 1.0)
    p = str(tmp_path / "legacy.t7")
    t7_writer.save_legacy(p, flat, past_flow, versioned=versioned, replicas=replicas)
    got, pf = load_t7(p)
    assert pf == past_flow
    np.testing.assert_array_equal(got, flat)
    # one shared storage: the file is about the size of the parameters, not twice
    assert os.path.getsize(p) < 1.6 * 4 * flat.size


def test_function_tags_are_framed_independently(tmp_path):
    """A table holding functions of all three tags followed by a marker: a reader that mis-frames one of them loses the
    marker (tag 6 has NO reference index: ADVICE r1)."""
    p = tmp_path / "f.t7"
    with open(p, "wb") as f:
        w = t7_writer.Writer(f)
        w.obj({"a": t7_writer.Function(6, {"x": 1}), "b": t7_writer.Function(8, {"y": 2}), "c": t7_writer.Function(7),
               "model": t7_writer.build_model(W.random_init(2, False, 1.0), False, cuda=False)})
    got, pf = load_t7(str(p))
    np.testing.assert_array_equal(got, W.random_init(2, False, 1.0))


def load_t7_ex(path, graph=None):
    L = _lib.lib()
    n = C.c_longlong()
    opts = C.create_string_buffer(256)
    g = graph.encode() if graph is not None else None
    _lib.check(L.b2f_load_t7_ex(path.encode(), g, None, 0, C.byref(n), opts, 256))
    out = np.empty(n.value, np.float32)
    _lib.check(L.b2f_load_t7_ex(path.encode(), g, _lib.fptr(out), out.size, C.byref(n), opts, 256))
    return out, dict(kv.split("=") for kv in opts.value.decode().split(","))


@pytest.mark.parametrize("win,levels,skip,past", [(5, 4, 2, True), (7, 5, 1, False), (3, 6, 3, True), (9, 7, 2, False), (3, 4, 0, True), (5, 2, 0, False)])
def test_other_graph_shapes_are_read_from_the_file(tmp_path, win, levels, skip, past):
    """createModelMulti(opt) with another window / number of levels / pwc_skip saved as .t7: the reader takes win from the
    nn.CostVolMulti nodes' win field (CostVolMulti.lua:26-37), levels from the convUnits and skip from the decoder levels in the
    node list (pwc.lua:136,237) and returns the weights in THAT graph's canonical order."""
    o = W.graph_opts(win=win, levels=levels, skip=skip)
    flat = W.random_init(17, past, 1.0, o)
    p = str(tmp_path / "m.t7")
    with open(p, "wb") as f:
        t7_writer.Writer(f).obj(t7_writer.build_model(flat, past, cuda=False, cudnn=False, o=o))
    got, opts = load_t7_ex(p)
    assert (int(opts["win"]), int(opts["levels"]), int(opts["skip"]), int(opts["past_flow"])) == (win, levels, skip, int(past))
    np.testing.assert_array_equal(got, flat)
    # given explicitly, the options must describe the file
    got2, _ = load_t7_ex(p, "win=%d,levels=%d,skip=%d" % (win, levels, skip))
    np.testing.assert_array_equal(got2, flat)
    with pytest.raises(_lib.B2FError, match="window|levels|decoder"):
        load_t7_ex(p, "win=%d,levels=%d,skip=%d" % (win + 2 if win < 13 else 3, levels, skip))
    if levels > 2:
        with pytest.raises(_lib.B2FError, match="levels"):
            load_t7_ex(p, "win=%d,levels=%d,skip=%d" % (win, levels - 1, min(skip, levels - 2)))
    if skip == 0:   # pwc_skip = 0 files carry a level-1 convUnit (pwc.lua:171-173); levels 1 and 2 are told apart by the node order
        with pytest.raises(_lib.B2FError, match="level-1 convUnit"):
            load_t7_ex(p, "win=%d,levels=%d,skip=1" % (win, levels))
    # b2f_load_t7 (no options) stays the loader of the shipped shape
    if (win, levels, skip) == (9, 7, 2):
        np.testing.assert_array_equal(load_t7(p)[0], flat)
    else:
        with pytest.raises(_lib.B2FError):
            load_t7(p)


@pytest.mark.parametrize("past", [False, True])
def test_skip0_file_written_back_to_front(tmp_path, past):
    """pwc_skip = 0 (pwc.lua:120-122): levels 1 and 2 both carry 16 maps, so their decoders have the same shapes and the reader
    tells them apart by the node order -- also in a file whose forward nodes are stored in reverse."""
    o = W.graph_opts(win=3, levels=4, skip=0)
    flat = W.random_init(23, past, 1.0, o)
    p = str(tmp_path / "m.t7")
    t7_writer.save_legacy(p, flat, past, o=o)
    got, opts = load_t7_ex(p)
    assert (int(opts["win"]), int(opts["levels"]), int(opts["skip"]), int(opts["past_flow"])) == (3, 4, 0, int(past))
    np.testing.assert_array_equal(got, flat)
