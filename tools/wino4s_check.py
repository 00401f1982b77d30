"""F(4x4) kernel on the bf16 matrix pipe with split operands (conv3x3_wino4s) against the fp32-MFMA F(4x4) kernel and an fp64
convolution through the conv3x3 op entry point (GPU box only): random shapes with ragged edges, odd / even chunk counts, one to
many tiles per block.  Prints the error of BOTH kernels against fp64: the split kernel must stay inside the fp32 kernel's bars.
    python tools/wino4s_check.py [seed] [cases]"""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from back2future_amd import back2future, ops

m = back2future.Model("random:hard:1:1.0")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
hyb = int(sys.argv[3]) if len(sys.argv) > 3 else 0      # > 0: check the hybrid form (wino4_hybrid = that many bf16 steps) instead of wino4_split
worst = [0.0, 0.0]
mean = [0.0, 0.0]
bad = 0
for it in range(n):
    ci = int(rng.choice([32, 40, 64, 96, 104, 128, 200]))
    co = int(rng.choice([64, 96, 100, 128, 160, 192]))
    h, w = int(rng.integers(1, 70)), int(rng.integers(1, 100))
    B = int(rng.integers(1, 4))
    scale = float(rng.choice([1.0, 1.0, 1e-3, 50.0]))
    x = (rng.standard_normal((B, ci, h, w), dtype=np.float32) * np.float32(scale)).astype(np.float32)
    wt = (rng.standard_normal((co, ci, 3, 3), dtype=np.float32) / np.sqrt(9 * ci)).astype(np.float32)
    b = (rng.standard_normal(co, dtype=np.float32) * np.float32(scale)).astype(np.float32)
    leaky = bool(rng.integers(0, 2))
    m.set_option("wino4_min_pixels", 0)          # F(4x4) at every size
    G = int(rng.choice([2, 3, 5, 8, 17, 64]))
    m.set_option("wino4_persistent", G)
    opt, val = ("wino4_hybrid", hyb) if hyb > 0 else ("wino2_split", 1) if hyb < 0 else ("wino4_split", 1)   # hyb < 0: the F(2x2) split kernel
    m.set_option(opt, 0)
    f32 = ops.conv3x3(m, x, wt, b, 1, leaky)
    m.set_option(opt, val)
    got = ops.conv3x3(m, x, wt, b, 1, leaky)
    m.set_option("wino4_persistent", 1)
    got1 = ops.conv3x3(m, x, wt, b, 1, leaky)
    m.set_option(opt, 0)
    y = torch.nn.functional.conv2d(torch.from_numpy(x).double(), torch.from_numpy(wt).double(), torch.from_numpy(b).double(), padding=1)
    if leaky:
        y = torch.where(y > 0, y, 0.2 * y)
    exp = y.numpy()
    es, ef = np.abs(got - exp) / scale, np.abs(f32 - exp) / scale
    worst = [max(worst[0], float(es.max())), max(worst[1], float(ef.max()))]
    mean = [mean[0] + float(es.mean()) / n, mean[1] + float(ef.mean()) / n]
    same = np.array_equal(got, got1)
    ok = np.isfinite(got).all() and es.max() < 2e-4 and same
    bad += 0 if ok else 1
    print("%3d B%d %3d->%3d %2dx%2d leaky=%d G=%2d scale %-6g | split: max %.2e mean %.2e | fp32 kernel: max %.2e mean %.2e | grid-independent %s%s"
          % (it, B, ci, co, h, w, leaky, G, scale, es.max(), es.mean(), ef.max(), ef.mean(), same, "" if ok else "   <-- BAD"), flush=True)
print("worst |err| / scale vs fp64: split %.3g, fp32 kernel %.3g; mean: split %.3g, fp32 kernel %.3g; bad cases %d" % (worst[0], worst[1], mean[0], mean[1], bad))
assert bad == 0
