"""Flow / image I/O around computeFlow, mirroring what the reference's README does with
`image` and flowExtensions.lua (README.md:56-68): load PNG frames as 3 x H x W floats in [0,1],
write / read Middlebury .flo files, flow visualisation (xy2rgb), save the masks.

    im1 = flow_io.load_image('samples/frame_0009.png')
    flow, fwd_occ, bwd_occ = computeFlow(im1, im2, im3)
    flow_io.writeFLO('flow.flo', flow.astype('float32'))
"""
import builtins
import struct

import numpy as np

TAG_FLOAT = 202021.25   # flowExtensions.lua:277


def load_image(path):
    """image.load(path) [3P]: C x H x W float in [0,1] (8-bit PNG / 255, 16-bit / 65535)."""
    from PIL import Image
    im = Image.open(path)
    a = np.asarray(im)
    scale = 65535.0 if a.dtype == np.uint16 else 255.0
    if a.ndim == 2:
        a = a[:, :, None]
    a = a[:, :, :3] if a.shape[2] >= 3 else np.repeat(a[:, :, :1], 3, 2)
    return np.ascontiguousarray(a.transpose(2, 0, 1)).astype(np.float32) / np.float32(scale)


def save_mask(path, mask):
    """image.save(path, mask * 255) for a 1 x H x W byte mask (README.md:66-67)."""
    from PIL import Image
    Image.fromarray((np.asarray(mask)[0] * 255).astype(np.uint8)).save(path)


def save_image(path, rgb):
    """image.save(path, img) [3P] for a 3 x H x W float image in [0,1] (README.md:63)."""
    from PIL import Image
    a = np.clip(np.asarray(rgb, np.float64), 0.0, 1.0)
    Image.fromarray((a.transpose(1, 2, 0) * 255.0 + 0.5).astype(np.uint8)).save(path)


def writeFLO(filename, F):
    """flowExtensions.lua:275-287: float tag 202021.25, int32 width, int32 height, then the
    2 x H x W flow interleaved as H x W x (u, v) float32, row major."""
    F = np.asarray(F, dtype=np.float32)
    assert F.ndim == 3 and F.shape[0] == 2
    with open(filename, "wb") as f:
        f.write(struct.pack("<f", TAG_FLOAT))
        f.write(struct.pack("<ii", F.shape[2], F.shape[1]))
        f.write(np.ascontiguousarray(F.transpose(1, 2, 0)).tobytes())


def loadFLO(filename):
    """flowExtensions.lua:254-273: returns the 2 x H x W float32 flow."""
    with open(filename, "rb") as f:
        tag = struct.unpack("<f", f.read(4))[0]
        if tag != TAG_FLOAT:
            raise ValueError("unable to read %s: wrong tag (big endian?)" % filename)
        w, h = struct.unpack("<ii", f.read(8))
        data = np.frombuffer(f.read(4 * 2 * w * h), dtype="<f4").reshape(h, w, 2)
    return np.ascontiguousarray(data.transpose(2, 0, 1))


# ---- flow visualisation of flowExtensions.lua:8-148 (README.md:61-63: flowX.xy2rgb(flow[1], flow[2])) ----
def computeNorm(flow_x, flow_y):
    """flowExtensions.lua:8-21: sqrt(x^2 + y^2)."""
    x, y = np.asarray(flow_x, np.float64), np.asarray(flow_y, np.float64)
    return np.sqrt(y * y + x * x)


def computeAngle(flow_x, flow_y):
    """flowExtensions.lua:23-51: direction in degrees, 0..360 (atan(|y/x|) folded into the quadrant;
    x == 0: 90 for y >= 0, 270 otherwise)."""
    x, y = np.asarray(flow_x, np.float64), np.asarray(flow_y, np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        h = np.degrees(np.arctan(np.abs(y / x)))
    out = np.where((x >= 0) & (y >= 0), h, 0.0)
    out = np.where((x >= 0) & (y < 0), 360.0 - h, out)
    out = np.where((x < 0) & (y >= 0), 180.0 - h, out)
    out = np.where((x < 0) & (y < 0), 180.0 + h, out)
    out = np.where((x == 0) & (y <= 0), 270.0, out)
    out = np.where((x == 0) & (y >= 0), 90.0, out)       # the Lua map2 tests this case first
    return out


def hsl2rgb(hsl):
    """image.hsl2rgb [3P, torch/image]: h, s, l in [0,1] -> r, g, b in [0,1]."""
    h, s, l = (np.asarray(hsl[i], np.float64) for i in range(3))
    q = np.where(l < 0.5, l * (1 + s), l + s - l * s)
    p = 2 * l - q

    def hue(t):
        t = np.where(t < 0, t + 1, t)
        t = np.where(t > 1, t - 1, t)
        return np.where(t < 1 / 6, p + (q - p) * 6 * t,
                        np.where(t < 1 / 2, q, np.where(t < 2 / 3, p + (q - p) * (2 / 3 - t) * 6, p)))
    rgb = np.stack([hue(h + 1 / 3), hue(h), hue(h - 1 / 3)])
    return np.where(s[None] == 0, l[None], rgb)


def field2rgb(norm, angle, max=None):
    """flowExtensions.lua:53-121 without the legend: hue = angle / 360, saturation = norm / max (tanh'ed when
    max is given), lightness = 1 - saturation / 2 (null flow is white).  Returns (rgb 3 x H x W, max)."""
    norm, angle = np.asarray(norm, np.float64), np.asarray(angle, np.float64)
    saturate = max is not None
    mx = builtins.max(max if max is not None else float(norm.max()), 1e-2)   # `max` is the reference's argument name
    s = norm / mx
    if saturate:
        s = np.tanh(s)
    return hsl2rgb(np.stack([angle / 360.0, s, 1.0 - 0.5 * s])), mx


def xy2rgb(x, y, max=None):
    """flowExtensions.lua:123-148: RGB picture of a flow field (saturation = magnitude, hue = direction)."""
    return field2rgb(computeNorm(x, y), computeAngle(x, y), max)

