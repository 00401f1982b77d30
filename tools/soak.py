"""Soak test of the host-buffer entry points on the GPU box: random shapes (also not multiples of 64), batch sizes,
input kinds (floats, k/255 floats, bytes; pageable or page-locked), sub-batch settings and graph replay on/off, each
result checked against the same triplets computed one at a time, device memory watched for growth.
    python tools/soak.py [iterations] [seed] [max H] [max W] [max n]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from back2future_amd import back2future  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    r = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    maxh = int(sys.argv[3]) if len(sys.argv) > 3 else 300
    maxw = int(sys.argv[4]) if len(sys.argv) > 4 else 420
    maxn = int(sys.argv[5]) if len(sys.argv) > 5 else 6
    models = {"hard": back2future.Model("random:hard:3:2.0"), "soft": back2future.Model("random:soft:3:2.0")}
    for m in models.values():
        m.set_option("wino4_min_pixels", 4096)
        m.set_option("adaptive_kernels", 0)          # kernel choice independent of the batch: bit-identical results (the default -1 picks per launch for single-triplet calls)
    free0 = None
    t0 = time.time()
    for it in range(iters):
        m = models["hard" if r.integers(2) else "soft"]
        H0, W0 = int(r.integers(64, maxh)), int(r.integers(64, maxw))
        n = int(r.integers(1, maxn + 1))
        kind = int(r.integers(3))                   # 0 floats, 1 k/255 floats, 2 bytes
        pinned = bool(r.integers(2))
        # options live in the context (the environment only seeds them in b2f_init): set them per iteration through the API
        sub_px, thr = int(r.choice([H0 * W0, 3 * H0 * W0, 1 << 22, 1 << 24])), int(r.choice([2, 5, 16]))
        m.set_option("host_subbatch_pixels", sub_px)
        m.set_option("host_threads", thr)
        m.set_option("host_graph", int(r.integers(2)))
        by = r.integers(0, 256, (3, n, 3, H0, W0), dtype=np.uint8)
        if kind == 2:
            ims = [by[i] for i in range(3)]
        elif kind == 1:
            ims = [(by[i].astype(np.float32) / np.float32(255.0)).astype(np.float32) for i in range(3)]
        else:
            ims = [r.random((n, 3, H0, W0), dtype=np.float32) for _ in range(3)]
        if pinned:
            keep = [torch.from_numpy(a).pin_memory() for a in ims]
            ims = [t.numpy() for t in keep]
        flow, fo, bo = m.computeFlowBatch(*ims)
        assert np.isfinite(flow).all()
        for i in set(int(x) for x in r.integers(0, n, 2)):
            one = [a[i:i + 1] for a in ims]
            f1, fo1, bo1 = m.computeFlowBatch(*one)
            if not (np.array_equal(f1[0], flow[i]) and np.array_equal(fo1[0], fo[i]) and np.array_equal(bo1[0], bo[i])):
                d = np.abs(f1[0] - flow[i])
                print("MISMATCH it %d %dx%d n=%d kind=%d pinned=%d triplet %d: max|dflow| %.3g at %s (%d values differ), masks differ %d / %d; env %s graph %s"
                      % (it, H0, W0, n, kind, pinned, i, d.max(), np.unravel_index(d.argmax(), d.shape), int((d > 0).sum()),
                         int((fo1[0] != fo[i]).sum()), int((bo1[0] != bo[i]).sum()),
                         {"host_subbatch_pixels": sub_px, "host_threads": thr}, m.get_option("host_graph")), flush=True)
                again = m.computeFlowBatch(*ims)
                print("   batch recomputed equals first batch result:", np.array_equal(again[0], flow), " single recomputed equals single:",
                      np.array_equal(m.computeFlowBatch(*one)[0], f1), flush=True)
                raise SystemExit(1)
        if it == 20:
            free0 = torch.cuda.mem_get_info()[0]
        if it % 25 == 0:
            print("it %d  %dx%d n=%d kind=%d pinned=%d  free %.2f GB  %.0f s" % (it, H0, W0, n, kind, pinned,
                  torch.cuda.mem_get_info()[0] / 2**30, time.time() - t0), flush=True)
    free1 = torch.cuda.mem_get_info()[0]
    print("done: %d iterations, device memory after warm-up %.2f GB free -> %.2f GB free" % (iters, free0 / 2**30, free1 / 2**30))
    if len(sys.argv) <= 3:      # with the default (small) shapes the buffers reach their final size within the warm-up
        assert free0 - free1 < (2 << 30), "device memory keeps growing"
    for m in models.values():
        m.close()


if __name__ == "__main__":
    main()
