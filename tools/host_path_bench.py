"""PCIe-inclusive rate of the host-buffer entry point (b2f_compute_flow_batch behind computeFlowBatch): caller
buffers in host memory in, f64 flow + u8 masks in host memory out.  Not bench.py's `value` (that one starts
with the inputs resident in HBM); DESIGN.md section 6 quotes these numbers next to it.

    python tools/host_path_bench.py [--n 32] [--height 1024] [--width 1920] [--reps 3]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from back2future_amd import back2future  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=32)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--u8", type=int, default=0, help="1: inputs are k / 255 (8-bit image data), eligible for the byte transport")
    ap.add_argument("--drop-symbol", action="append", default=[], help="for A/B runs against an older B2F_LIB build")
    a = ap.parse_args()
    from back2future_amd import _lib
    for sym in a.drop_symbol:
        _lib.SIGNATURES.pop(sym)
    n, H, W = a.n, a.height, a.width
    m = back2future.Model("random:soft:5:2.0")
    g = torch.Generator().manual_seed(1)
    ims = [torch.rand((n, 3, H, W), generator=g) for _ in range(3)]
    if a.u8:
        ims = [torch.round(t * 255.0) / 255.0 for t in ims]
    res = {"n": n, "H": H, "W": W, "u8_data": a.u8, "env": {k: v for k, v in os.environ.items() if k.startswith("B2F_")}}

    def timed(fn):
        fn()                                   # buffers, page faults
        fn()                                   # hipGraph capture (second use of a shape)
        best = 1e30
        for _ in range(a.reps):
            t = time.perf_counter()
            fn()
            best = min(best, time.perf_counter() - t)
        return best

    # 1. pageable caller buffers (numpy): staged through the library's pinned sets by host threads
    pg = [t.numpy() for t in ims]
    out_pg = (np.empty((n, 2, H, W), np.float64), np.empty((n, 1, H, W), np.uint8), np.empty((n, 1, H, W), np.uint8))
    t = timed(lambda: m.computeFlowBatch(*pg, out=out_pg))
    res["pageable_triplets_per_s"] = n / t
    # 2. page-locked caller buffers: DMA in place
    pin = [t_.pin_memory() for t_ in ims]
    out_pin = (torch.empty((n, 2, H, W), dtype=torch.float64).pin_memory(),
               torch.empty((n, 1, H, W), dtype=torch.uint8).pin_memory(),
               torch.empty((n, 1, H, W), dtype=torch.uint8).pin_memory())
    pin_np = [t_.numpy() for t_ in pin]
    out_np = tuple(t_.numpy() for t_ in out_pin)
    t = timed(lambda: m.computeFlowBatch(*pin_np, out=out_np))
    res["pinned_triplets_per_s"] = n / t
    assert np.array_equal(out_np[0], out_pg[0]) and np.array_equal(out_np[1], out_pg[1])
    if a.u8:   # 2b. the byte entry point (frames still 8-bit): no packing pass on the host
        by = [torch.round(t_ * 255.0).to(torch.uint8).numpy() for t_ in ims]
        t = timed(lambda: m.computeFlowBatch(*by, out=out_pg))
        res["bytes_in_triplets_per_s"] = n / t
        assert np.array_equal(out_np[0], out_pg[0])
        one_b = [b_[0:1] for b_ in by]
        o1b = tuple(o[0:1] for o in out_pg)
        res["single_triplet_ms_bytes_in"] = timed(lambda: m.computeFlowBatch(*one_b, out=o1b)) * 1e3
    # 3. one triplet at a time (the reference's computeFlow signature): latency
    one = [p[0] for p in pg]
    t = timed(lambda: m.computeFlow(*one))
    res["single_triplet_ms_pageable"] = t * 1e3
    one_pin = [p[0:1] for p in pin_np]
    o1 = tuple(o[0:1] for o in out_np)
    t = timed(lambda: m.computeFlowBatch(*one_pin, out=o1))
    res["single_triplet_ms_pinned"] = t * 1e3
    # bytes over PCIe per triplet
    res["h2d_mb_per_triplet"] = 36 * H * W / 1e6
    res["d2h_mb_per_triplet"] = 18 * H * W / 1e6
    print(json.dumps(res))
    m.close()


if __name__ == "__main__":
    main()
