"""Two contexts driven by two host threads on the same GPU (the reference runs one Lua thread per GPU replica,
util.lua:34-40): every result must equal the single-threaded one bit for bit.   python tools/two_contexts_check.py"""
import os, sys, threading
import numpy as np
sys.path.insert(0, '.')
import torch
from back2future_amd import back2future
os.environ["B2F_WINO4_MIN_PIXELS"] = "4096"
r = np.random.default_rng(3)
jobs = [(int(r.integers(1, 5)), int(r.integers(64, 260)), int(r.integers(64, 400))) for _ in range(40)]
data = [[r.random((n, 3, H, W), dtype=np.float32) for _ in range(3)] for (n, H, W) in jobs]
ref_m = back2future.Model("random:soft:4:2.0")
ref = [ref_m.computeFlowBatch(*d) for d in data]
ref_m.close()
errs = []
def worker(tid):
    m = back2future.Model("random:soft:4:2.0")
    for rep in range(3):
        for k in range(tid, len(jobs), 2) if rep % 2 == 0 else range(len(jobs) - 1 - tid, -1, -2):
            out = m.computeFlowBatch(*data[k])
            for a, b in zip(out, ref[k]):
                if not np.array_equal(a, b): errs.append((tid, rep, k))
    m.close()
th = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
[t.start() for t in th]; [t.join() for t in th]
print("two contexts on two threads:", "OK" if not errs else errs[:5])
