"""CPU, world_size 2, gloo: the multi-rank host logic of the path -- static sharding of the
triplets and the single weight broadcast -- without a GPU."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from back2future_amd import dist as D, weights as W


def test_shard_range_partitions():
    for n in (0, 1, 7, 8, 16, 127, 128):
        for world in (1, 2, 3, 8):
            spans = [D.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert D.shard_range(128, 3, 8) == (48, 64)       # BASELINE configs[4]: 16 triplets per GPU


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = W.param_count(False)
        flat = torch.from_numpy(W.random_init(2, False, 1.0)) if rank == 0 else torch.zeros(n)
        D.broadcast_flat(flat, src=0)
        ok_w = bool(np.array_equal(flat.numpy(), W.random_init(2, False, 1.0)))
        # shard 5 "triplets", process locally (stand-in: per-triplet checksum), gather on every rank
        lo, hi = D.shard_range(5, rank, world)
        data = np.arange(5 * 4, dtype=np.float32).reshape(5, 4)
        local = data[lo:hi].sum(1, keepdims=True)
        allv = D.gather_host(local)
        ok_g = bool(np.array_equal(allv, data.sum(1, keepdims=True)))
        q.put((rank, ok_w, ok_g, hi - lo))
    finally:
        dist.destroy_process_group()


def test_broadcast_and_shard_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [r[0] for r in res] == [0, 1]
    assert all(r[1] and r[2] for r in res)
    assert sum(r[3] for r in res) == 5


def test_bench_plain_invocation_starts_the_ranks_itself():
    """`python bench.py --gpus 2` with no launcher: the parent (which never imports torch or touches a GPU) starts both
    ranks and hands back their exit code.  Without a GPU every rank refuses loudly (no CPU fallback), so here rc = 1 and
    both refusals are visible; the GPU twin is tests/test_gpu_bench.py::test_plain_invocation_spawns_ranks."""
    import subprocess
    import sys
    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpu"],
                       capture_output=True, text=True, cwd=root, env=env, timeout=600)
    assert r.returncode == 1
    assert r.stderr.count("bench.py needs a GPU") == 2 and "ranks failed" in r.stderr
