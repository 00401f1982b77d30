"""CPU: .flo round trip (flowExtensions.lua:254-287 wire format) and PNG loading."""
import struct

import numpy as np

from back2future_amd import flow_io


def test_flo_roundtrip_and_layout(tmp_path):
    rng = np.random.default_rng(0)
    F = rng.standard_normal((2, 5, 7)).astype(np.float32)
    p = str(tmp_path / "f.flo")
    flow_io.writeFLO(p, F)
    raw = open(p, "rb").read()
    assert struct.unpack("<f", raw[:4])[0] == 202021.25
    assert struct.unpack("<ii", raw[4:12]) == (7, 5)                      # width first, then height
    first = np.frombuffer(raw[12:20], "<f4")
    assert first[0] == F[0, 0, 0] and first[1] == F[1, 0, 0]              # interleaved (u, v)
    assert len(raw) == 12 + 4 * 2 * 5 * 7
    np.testing.assert_array_equal(flow_io.loadFLO(p), F)


def test_load_image(tmp_path):
    from PIL import Image
    a = (np.arange(4 * 6 * 3) % 256).astype(np.uint8).reshape(4, 6, 3)
    p = str(tmp_path / "x.png")
    Image.fromarray(a).save(p)
    im = flow_io.load_image(p)
    assert im.shape == (3, 4, 6) and im.dtype == np.float32
    np.testing.assert_allclose(im, a.transpose(2, 0, 1) / 255.0, rtol=1e-6)
