"""GPU box: what ONE process's host pipeline can feed (b2f_compute_flow_batch_u8: pageable byte frames in, f64 flow + u8 masks out) as a function
of its copy threads (option host_threads) -- the number DESIGN.md section 7 needs to say whether the one-process multi-GPU entry point scales past
a GPU or two when the frames live in pageable host memory.
    python tools/host_staging_rate.py [n] [H W]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from back2future_amd import back2future

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
H, W = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1024, 1920)
r = np.random.default_rng(0)
by = [r.integers(0, 256, (n, 3, H, W), dtype=np.uint8) for _ in range(3)]
m = back2future.Model("random:hard:2:1.0")
mb_in = 9.0 * H * W / 1e6
mb_out = (2 * 8 + 2) * H * W / 1e6
print("per triplet: %.1f MB of byte frames in, %.1f MB out (f64 flow + two u8 masks); host cores: %d" % (mb_in, mb_out, os.cpu_count()))
out = (np.empty((n, 2, H, W), np.float64), np.empty((n, 1, H, W), np.uint8), np.empty((n, 1, H, W), np.uint8))   # the caller's buffers, reused (a fresh 1-GB
                                                                                                                 # array per call is page faults, not staging)
for th in (2, 3, 4, 6, 8, 12, 16, 24, 32):
    m.set_option("host_threads", th)
    m.computeFlowBatch(*by, out=out)
    m.computeFlowBatch(*by, out=out)
    reps, dt = 3, 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        m.computeFlowBatch(*by, out=out)
        dt = min(dt, time.perf_counter() - t0)
    print("host_threads %2d: %7.1f triplets/s  (%.1f GB/s in + %.1f GB/s out through host memory)" % (th, n / dt, n * mb_in / dt / 1e3, n * mb_out / dt / 1e3), flush=True)
m.close()
