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
