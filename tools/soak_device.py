"""Soak test of the device-pointer entry points on the GPU box: random batch sizes and shapes in random order on one
context (so the workspace arena, the graph cache and the per-shape plans keep changing), on torch's current stream or
a side stream, eager or graph replay; every pass is checked against the same triplets computed one at a time, and
now and then against a fresh context and the full-table entry point.
    python tools/soak_device.py [iterations] [seed] [big]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from back2future_amd import back2future  # noqa: E402


def run(m, x, stream=None, unit=True):
    B, _, H, W = x.shape
    flow = torch.empty(B, 2, H, W, device="cuda"); occ = torch.empty(B, 2, H, W, device="cuda")
    est3 = torch.empty(B, 2 if m.past_flow else 3, H, W, device="cuda")
    # stream None = the context's own (non-blocking) stream, waited for with b2f_synchronize
    m.forward_device(x.data_ptr(), B, H, W, flow.data_ptr(), occ.data_ptr(), est3.data_ptr(), unit_input=unit,
                     stream=stream.cuda_stream if stream is not None else None)
    if stream is not None:
        stream.synchronize()
    else:
        m.synchronize()
    return flow, occ, est3


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    big = len(sys.argv) > 3 and sys.argv[3] == "big"   # every fourth pass at 384..704 x 512..1024
    r = np.random.default_rng(seed)
    os.environ["B2F_WINO4_MIN_PIXELS"] = "4096"      # kernel choice independent of the batch: bit-identical results
    g = torch.Generator(device="cuda").manual_seed(seed)
    models = {"hard": back2future.Model("random:hard:3:2.0"), "soft": back2future.Model("random:soft:3:2.0")}
    for m in models.values():
        m.set_option("adaptive_kernels", 0)          # kernel choice by map size also for the single-triplet reference passes (the default -1 picks per launch there: 1e-6-level differences)
    side = torch.cuda.Stream()
    t0 = time.time()
    for it in range(iters):
        which = "hard" if r.integers(2) else "soft"
        m = models[which]
        B = int(r.integers(1, 7))
        H, W = 64 * int(r.integers(1, 6)), 64 * int(r.integers(1, 8))
        if big and it % 4 == 0: H, W = 64 * int(r.integers(6, 12)), 64 * int(r.integers(8, 17))   # level-3 maps of >= 16 384 pixels: the F(6x6) kernel
        m_graph = int(r.integers(2))
        m.set_option("use_graph", m_graph)
        stream = side if r.integers(2) else None
        x = torch.rand((B, 9, H, W), generator=g, device="cuda")
        torch.cuda.synchronize()                       # inputs are produced on torch's stream, consumed on another
        outs = run(m, x, stream)
        assert all(bool(torch.isfinite(o).all()) for o in outs)
        for i in set(int(v) for v in r.integers(0, B, 2)):
            one = run(m, x[i:i + 1].contiguous(), stream)
            for name, a, b in zip(("flow", "occ", "est3"), one, outs):
                if not torch.equal(a[0], b[i]):
                    d = (a[0] - b[i]).abs()
                    print("MISMATCH it %d %s B=%d %dx%d triplet %d %s: max %.3g, %d values, graph=%s side=%s" % (
                        it, which, B, H, W, i, name, float(d.max()), int((d > 0).sum()), m_graph, stream is not None), flush=True)
                    again = run(m, x, stream)
                    print("   batch again equals batch:", [bool(torch.equal(p_, q_)) for p_, q_ in zip(again, outs)], flush=True)
                    raise SystemExit(1)
        if it % 40 == 7:      # a fresh context must give the same bits as the long-lived one
            fresh = back2future.Model("random:%s:3:2.0" % which)
            for a, b in zip(run(fresh, x), outs):
                assert torch.equal(a, b), (it, "fresh context differs")
            fresh.close()
        if it % 40 == 23 and B <= 2 and H * W <= 192 * 256:      # full output table: est[1] is the same flow
            mean = torch.tensor([0.485, 0.456, 0.406] * 3, device="cuda").view(1, 9, 1, 1)
            std = torch.tensor([0.229, 0.224, 0.225] * 3, device="cuda").view(1, 9, 1, 1)
            xn = ((x + (-mean)) / std).cpu().numpy()
            table = m.forward(xn)
            assert np.array_equal(table[0], outs[0].cpu().numpy()), (it, "full table flow differs")
        if it % 50 == 0:
            print("it %d %s B=%d %dx%d  free %.2f GB  %.0f s" % (it, which, B, H, W, torch.cuda.mem_get_info()[0] / 2**30, time.time() - t0), flush=True)
    print("done: %d iterations" % iters)
    for m in models.values():
        m.close()


if __name__ == "__main__":
    main()
