#!/usr/bin/env python3
"""The reference's README example (README.md:45-68) on the MI355X path:

    back2future = require('back2future'); computeFlow = back2future.init('Ours-Soft-ft-KITTI')
    flow, fwd_occ, bwd_occ = computeFlow(im1, im2, im3); flowX.writeFLO(...); image.save(...)

Usage: python examples/run_samples.py [model] [frame1 frame2 frame3] [out_prefix]
model: 'Ours-Hard' | 'Ours-Soft-ft-KITTI' | 'Ours-Soft-ft-Sintel' (needs models/RoamingImages_*.t7 in the
current directory, as in the reference) or 'random:soft' / a .t7 / .b2fw path.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from back2future_amd import back2future, flow_io   # noqa: E402


def main():
    args = sys.argv[1:]
    model = args[0] if args else "random:soft"
    s = os.path.join(ROOT, "tests", "golden", "samples")
    frames = args[1:4] if len(args) >= 4 else [os.path.join(s, "frame_%04d.png" % i) for i in (9, 10, 11)]
    prefix = args[4] if len(args) >= 5 else "flow"
    computeFlow = back2future.init(model)
    im1, im2, im3 = [flow_io.load_image(f) for f in frames]
    flow, fwd_occ, bwd_occ = computeFlow(im1, im2, im3)
    flow_io.writeFLO(prefix + ".flo", flow.astype("float32"))
    rgb, _ = flow_io.xy2rgb(flow[0], flow[1])                       # README.md:61-63
    flow_io.save_image(prefix + ".png", rgb)
    flow_io.save_mask(prefix + "_fwd_occ.png", fwd_occ)
    flow_io.save_mask(prefix + "_bwd_occ.png", bwd_occ)
    print("flow", flow.shape, "range", float(flow.min()), float(flow.max()),
          "fwd_occ", int(fwd_occ.sum()), "bwd_occ", int(bwd_occ.sum()))


if __name__ == "__main__":
    main()
