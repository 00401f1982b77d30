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


def test_xy2rgb_known_answers():
    """flowExtensions.lua:8-148: zero flow is white, full-magnitude +x flow is pure red (hue 0, s 1, l 0.5),
    +y is hue 90 deg, -x hue 180 deg (cyan), angle quadrants as in computeAngle."""
    x = np.array([[0.0, 2.0, 0.0, -2.0, 1.0, 1.0, -1.0]])
    y = np.array([[0.0, 0.0, 2.0, 0.0, 1.0, -1.0, -1.0]])
    ang = flow_io.computeAngle(x, y)
    np.testing.assert_allclose(ang, [[90.0, 0.0, 90.0, 180.0, 45.0, 315.0, 225.0]])
    np.testing.assert_allclose(flow_io.computeNorm(x, y)[0, :4], [0, 2, 2, 2])
    rgb, mx = flow_io.xy2rgb(x, y)
    assert mx == 2.0 and rgb.shape == (3, 1, 7)
    np.testing.assert_allclose(rgb[:, 0, 0], [1, 1, 1])            # null flow: white
    np.testing.assert_allclose(rgb[:, 0, 1], [1, 0, 0], atol=1e-12)  # +x at max: red
    np.testing.assert_allclose(rgb[:, 0, 3], [0, 1, 1], atol=1e-12)  # -x at max: cyan
    np.testing.assert_allclose(rgb[:, 0, 2], [0.5, 1, 0], atol=1e-12)  # hue 90 deg
    rgb2, mx2 = flow_io.xy2rgb(x, y, max=4.0)                       # explicit max: tanh saturation
    assert mx2 == 4.0
    s = np.tanh(0.5)
    l = 1 - 0.5 * s
    chroma = (1 - abs(2 * l - 1)) * s                               # HSL at hue 0: r = l + C/2, g = b = l - C/2
    np.testing.assert_allclose(rgb2[:, 0, 1], [l + chroma / 2, l - chroma / 2, l - chroma / 2], atol=1e-12)
