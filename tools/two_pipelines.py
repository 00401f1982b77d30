"""Throughput of P step pipelines in flight on one GPU (GPU box only): P contexts (own arena, own stream, own captured graph) take
the steps in turn, so the launch-latency-bound coarse levels of one step can run beside the wide kernels of another.
    python tools/two_pipelines.py [P ...]"""
import sys
import time
sys.path.insert(0, ".")
import torch
import bench
from back2future_amd import back2future

B, H, W, steps = 16, 1024, 1920, 20
dev = torch.device("cuda", 0)
for P in [int(a) for a in sys.argv[1:]] or [1, 2]:
    models = [back2future.Model("random:hard:2:1.0") for _ in range(P)]
    streams = [torch.cuda.Stream() for _ in range(P)]
    xs = [bench.make_triplets(torch, B, H, W, seed=2 + i, device=dev) for i in range(P)]
    outs = [(torch.empty(B, 2, H, W, device=dev), torch.empty(B, 2, H, W, device=dev), torch.empty(B, 3, H, W, device=dev)) for _ in range(P)]
    torch.cuda.synchronize()
    for m in models:
        m.set_option("use_graph", 1)

    def step(i):
        k = i % P
        f, o, e = outs[k]
        models[k].forward_device(xs[k].data_ptr(), B, H, W, f.data_ptr(), o.data_ptr(), e.data_ptr(), unit_input=True, stream=streams[k].cuda_stream)

    for i in range(3 * P):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("pipelines %d: %.1f triplets/s  (%.3f ms per step of %d triplets)" % (P, B * steps / dt, 1e3 * dt / steps, B), flush=True)
    for m in models:
        m.close()
    del models, xs, outs
    torch.cuda.empty_cache()
