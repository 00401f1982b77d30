"""Multi-GPU side of the computeFlow path: one process per GPU, triplets sharded statically
across ranks, ONE collective -- the broadcast of the flat weight buffer from rank 0 (RCCL over
xGMI when the backend is "nccl").  This replaces nn.DataParallelTable(1, true, true) of
util.lua:27-48 (scatter/gather of the batch + NCCL parameter sync, train.lua:494-496) for
inference: triplets are independent (computeFlow keeps no cross-sample state), so there is
no data-path collective.  Works with gloo on CPU tensors too (tests/test_dist_cpu.py)."""
import numpy as np


def shard_range(n, rank, world):
    """Static contiguous split of n triplets over `world` ranks (the batch split of
    DataParallelTable dim 1, util.lua:32): returns [lo, hi) owned by `rank`; sizes differ by
    at most one, earlier ranks take the remainder."""
    assert 0 <= rank < world and n >= 0
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def broadcast_flat(flat, src=0):
    """Broadcast a flat fp32 weight tensor (torch CPU or CUDA tensor) in place."""
    import torch.distributed as dist
    dist.broadcast(flat, src=src)
    return flat


def broadcast_weights(model, src=0):
    """RCCL-broadcast rank `src`'s weights into `model` (a back2future.Model) on every rank,
    in place on the library's own device buffer, then rebuild the packed kernel-side copies."""
    import torch
    import torch.distributed as dist
    ptr, n = model.weights_device_ptr()
    dev = torch.device("cuda", model.device)      # the GPU the context was created on, not torch's current one
    assert torch.cuda.current_device() == model.device, \
        "broadcast_weights: torch's current device (%d) is not the context's (%d); call torch.cuda.set_device first" \
        % (torch.cuda.current_device(), model.device)
    # wrap the library's device buffer without copying: broadcast straight into it
    class _Ext(object):
        pass
    holder = _Ext()
    holder.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False), "version": 2}
    flat = torch.as_tensor(holder, device=dev)
    dist.broadcast(flat, src=src)
    torch.cuda.synchronize(dev)
    model.commit_weights()
    return n


def weights_checksum(model):
    """64-bit checksum of the context's flat weight buffer as it sits on the device (after a broadcast every rank
    must report the same value)."""
    import zlib
    w = model.get_weights()
    return (zlib.crc32(w.tobytes()) << 32) | zlib.adler32(w.tobytes())


def gather_host(local, counts=None):
    """Concatenate per-rank numpy results on every rank (host side, after the timed path)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    objs = [None] * world
    dist.all_gather_object(objs, np.asarray(local))
    return np.concatenate(objs, 0)
