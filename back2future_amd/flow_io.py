"""Flow / image I/O around computeFlow, mirroring what the reference's README does with
`image` and flowExtensions.lua (README.md:56-68): load PNG frames as 3 x H x W floats in [0,1],
write / read Middlebury .flo files, save the masks.

    im1 = flow_io.load_image('samples/frame_0009.png')
    flow, fwd_occ, bwd_occ = computeFlow(im1, im2, im3)
    flow_io.writeFLO('flow.flo', flow.astype('float32'))
"""
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
