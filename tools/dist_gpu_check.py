# single-rank RCCL group on the 1-GPU box: exercises the device-pointer broadcast path of bench.py --gpus N
import os, sys
sys.path.insert(0, '.')
import numpy as np, torch, torch.distributed as dist
from back2future_amd import back2future, dist as bd
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
m = back2future.Model("random:hard:3:1.0", device=0)
w0 = m.get_weights().copy()
n = bd.broadcast_weights(m, src=0)
assert n == w0.size and np.array_equal(m.get_weights(), w0)
t = torch.tensor([1.5], device="cuda", dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.barrier()
print("dist path ok", n, float(t.item()))
dist.destroy_process_group()
