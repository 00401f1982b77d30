#!/usr/bin/env python3
"""bench.py -- frame-triplets/sec of the computeFlow hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: starts the N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One process per GPU.  A step = one pass of the whole computeFlow device pipeline
(pack+normalize -> siamese feature pyramid -> 5 x (fused warp + cost volume -> flow decoder)
-> occlusion decoder + softmax -> upsampling -> est[3]) over one batch of synthetic
3 x 1024 x 1920 triplets that already sit in HBM (BASELINE.json configs[4]: 16 triplets per
GPU).  Triplets are independent, so ranks share nothing in the timed region: the only
collective is the RCCL broadcast of the flat weight buffer at start-up (weak scaling).

Prints ONE JSON line on rank 0 (see the task contract); `roofline` is the dominant kernel
class (conv3x3 fp32-MFMA implicit GEMM), `roofline_corrwarp` the HBM-bound fused
warp + cost-volume kernel that BASELINE.json's target is quoted on; both are timed live with
HIP events on the launch stream in a second, un-timed eager pass of the same steps (the timed steps replay a hipGraph).  `cpu_baseline` times the CPU oracle
(oracle/, kind "port") on a bounded sample on rank 0 at N=1.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FEAT = [0, 3, 16, 32, 64, 96, 128, 192]
DEC = [128, 128, 96, 64, 32, 2]


def conv_flops_per_px():
    """Algorithmic conv FLOPs (2*MAC) per full-resolution pixel of the pruned computeFlow
    graph: features (3 frames) + flow decoders l=7..3 + occ decoder l=3."""
    feat = sum(3 * 18 * (FEAT[l - 1] * FEAT[l] + FEAT[l] ** 2) / 4 ** (l - 1) for l in range(2, 8))

    def dec(n):
        ci, s = n, 0
        for co in DEC:
            s += 18 * ci * co
            ci = co
        return s
    flow = sum(dec(162 if l == 7 else 162 + FEAT[l] + 2) / 4 ** (l - 1) for l in range(3, 8))
    occ3 = dec(162 + FEAT[3] + 2) / 4 ** 2
    return feat + flow + occ3


def conv_layers(H, W):
    """Every conv of the pruned computeFlow graph as (cin, cout, stride, out_h, out_w, calls per triplet, cin as the kernels see it):
    the three siamese towers, the flow decoders of levels 7..3 and the occlusion decoder of level 3.  Input
    channels are padded to 8; a decoder's first layer reads the 168-slot cost-volume record (flow inside) + the reference features."""
    out = []
    for l in range(2, 8):
        h, w = H >> (l - 1), W >> (l - 1)
        out.append((FEAT[l - 1], FEAT[l], 2, h, w, 3, (FEAT[l - 1] + 7) // 8 * 8))
        out.append((FEAT[l], FEAT[l], 1, h, w, 3, FEAT[l]))
    for l in range(7, 2, -1):
        h, w = H >> (l - 1), W >> (l - 1)
        ci = 162 if l == 7 else 162 + FEAT[l] + 2
        cip = 168 if l == 7 else 168 + FEAT[l]
        for co in DEC:
            out.append((ci, co, 1, h, w, 2 if l == 3 else 1, cip))
            ci = cip = co
    return out


MODE_CLASS = {"W6": "conv3x3_wino6", "W4": "conv3x3_wino4", "W2": "conv3x3_wino", "N2": "conv3x3_narrow2", "C16": "conv3x3_c16", "S16": "conv3x3_s2x16", "D1": "conv3x3_s1", "D2": "conv3x3_s2",
              "B16": "conv3x3_c16_bf16", "H16": "conv_head16_bf16", "E1": "conv3x3_s1_bf16", "E2": "conv3x3_s2_bf16", "V1": "conv3x3_w1b", "L2": "conv3x3_s2b"}
BF16_PIPE = ("conv3x3_c16_bf16", "conv_head16_bf16", "conv3x3_s1_bf16", "conv3x3_s2_bf16", "conv3x3_w1b", "conv3x3_s2b")   # split-operand kernels on the bf16 matrix pipe: no fp32-MFMA FLOPs


def layer_kernels(model, step, torch):
    """Kernel class of every conv layer AS THE LIBRARY RAN IT: one eager pass with option profile_layers, whose rows are named
    conv<mode>_<cin>to<cout>_<H>x<W> by b2f_api.hip:run_conv (mode W6 = Winograd F(6x6) [a last block of <= 32 outputs on the F(4x4) kernel], W4 = Winograd F(4x4), W2 = F(2x2), N2 = 2-output VALU kernel,
    C16 = 16 -> 16 kernel, S16 = 16 -> 32 stride-2 kernel, D1 / D2 = direct kernel stride 1 / 2, B16 = the 16 -> 16 layer on the bf16 pipe,
    E1 / E2 = direct kernel on the bf16 pipe stride 1 / 2, H16 = the fused head: 16 -> 16 and 16 -> 32 stride 2 in one kernel on the bf16 pipe, one row for both layers).
    Returns {(cin_padded, cout, H_in, W_in): class}."""
    model.set_option("use_graph", 0)
    model.set_option("profile_layers", 1)
    model.set_option("profile", 1)
    model.profile_reset()
    step()
    torch.cuda.synchronize()
    rows = model.profile_read()
    model.set_option("profile", 0)
    model.set_option("profile_layers", 0)
    model.profile_reset()
    import re
    out = {}
    for name, (ms, n) in rows.items():
        m = re.match(r"^conv(W6|W4|W2|N2|C16|S16|D1|D2|B16|H16|E1|E2|V1|L2)_(\d+)to(\d+)_(\d+)x(\d+)$", name)
        if m and n > 0:            # (rows of earlier passes keep their names with zero counts)
            out[(int(m.group(2)), int(m.group(3)), int(m.group(4)), int(m.group(5)))] = MODE_CLASS[m.group(1)]
            if m.group(1) == "H16":  # the fused head also holds the 16 -> 16 layer in front of its 16 -> 32 one
                out[(16, 16, int(m.group(4)), int(m.group(5)))] = MODE_CLASS["H16"]
    return out


def conv_flops_by_kernel(H, W, kernels):
    """Per kernel class and triplet: (algorithmic direct-convolution FLOPs, FLOPs the MFMA pipe executes); `kernels` = the
    library's own choice per layer (layer_kernels).  Executed: F(6x6) 64/36 MACs per output and channel pair, F(4x4) 36/16, F(2x2) 16/4, direct 9;
    input channels as conv_layers pads them, outputs to 32 (16 for the 16 -> 16 kernel); VALU kernels 0."""
    alg, exe = {}, {}
    for ci, co, stride, h, w, calls, cip in conv_layers(H, W):
        k = "conv_first" if ci == 3 else kernels[(cip, co, h * stride, w * stride)]
        n = float(h * w * calls)
        cop = (co + 31) // 32 * 32
        per_out = {"conv3x3_wino4": 36.0 / 16.0, "conv3x3_wino": 16.0 / 4.0}.get(k, 9.0)
        e = 2.0 * per_out * cip * cop
        if k == "conv3x3_wino6":   # blocks of 64 outputs with more than 32 real ones at 64/36 MACs per output, a last block of <= 32 outputs on F(4x4)
            c6 = 64 * (co // 64 + (1 if co % 64 > 32 else 0))
            e = 2.0 * cip * ((64.0 / 36.0) * min(c6, (co + 63) // 64 * 64) + (36.0 / 16.0) * max(0, cop - c6))
        if k in ("conv_first", "conv3x3_narrow2") or k in BF16_PIPE:
            e = 0.0
        if k == "conv3x3_c16":
            e = 2.0 * 9.0 * 16 * 16
        alg[k] = alg.get(k, 0.0) + 2.0 * 9.0 * ci * co * n
        exe[k] = exe.get(k, 0.0) + e * n
    return alg, exe


def corr_bytes_per_px():
    """Compulsory HBM bytes of the fused warp + cost volume per full-res pixel:
    sum_l (3 C_l + 2 [l<7] + 162) * 4 / 4^(l-1) = 97.17."""
    return sum((3 * FEAT[l] + (2 if l < 7 else 0) + 162) * 4 / 4 ** (l - 1) for l in range(3, 8))


def make_triplets(torch, B, H, W, seed, device):
    """im1 ~ U[0,1) smoothed by a 5x5 box, im2/im3 = im1 translated by (+3,+1)/(+6,+2) px plus
    U(-0.02,0.02) noise (SURVEY s8d 'Synthetic inputs'), planar B x 9 x H x W in [0,1]."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    base = torch.rand(B, 3, H + 16, W + 16, generator=g, device=device)
    base = torch.nn.functional.avg_pool2d(base, 5, stride=1, padding=2)
    out = torch.empty(B, 9, H, W, device=device)
    for f, (dx, dy) in enumerate([(0, 0), (3, 1), (6, 2)]):
        crop = base[:, :, 8 - dy:8 - dy + H, 8 - dx:8 - dx + W]
        noise = (torch.rand(B, 3, H, W, generator=g, device=device) - 0.5) * 0.04
        out[:, 3 * f:3 * f + 3] = (crop + noise).clamp_(0, 1)
    return out.contiguous()


def cpu_baseline(H, W, seed, n=2):
    """Both CPU columns time the SAME graph `value` runs -- the pruned computeFlow graph (365 GFLOP per 1024x1920 triplet)
    -- on a bounded sample of n triplets of the bench shape, on the GPU box's host cores:
      cpu_baseline        the oracle (oracle/b2f_oracle.c, kind 'port': literal restatement of the reference's loops,
                          scalar direct convolution, OpenMP over (image, output channel)) -- a correctness tool, slow by design;
      cpu_baseline_torch  the same graph in PyTorch-CPU (oneDNN convolutions, all threads): the closest stand-in for the
                          reference's own CPU route, Torch7 nn:float()."""
    import numpy as np
    import torch
    from back2future_amd import weights as Wt
    from oracle import oracle as O, torch_cpu as T
    rng = np.random.default_rng(seed)
    x = rng.random((n, 9, H, W), dtype=np.float32)
    params = Wt.random_init(2, False, 1.0)
    O.lib()
    t0 = time.perf_counter()
    table = O.pwc_forward(x, params, False, pruned=True)
    dt = time.perf_counter() - t0
    port = {"value": n / dt, "unit": "triplets/s", "cores": os.cpu_count(), "kind": "port",
            "sample": "%d triplets at 3x%dx%d, pruned computeFlow graph of the Ours-Hard shape (the graph `value` runs), "
                      "oracle/b2f_oracle.c with OpenMP on all host cores, %.2f s" % (n, H, W, dt)}
    # PyTorch's intra-op pool does not scale to every core of a 2-socket host for this graph (128 threads measured 3x
    # slower than 32): time a few pool sizes, report the best with the threads it used
    best = None
    ncpu = os.cpu_count() or 8
    for nthr in sorted({min(ncpu, n) for n in (16, 32, 64, 128)}):
        torch.set_num_threads(nthr)
        T.compute_flow_graph(x[:1], params, False)                   # untimed: oneDNN builds its primitives per shape / pool size
        t0 = time.perf_counter()
        flow, occ = T.compute_flow_graph(x, params, False)
        dt = time.perf_counter() - t0
        if best is None or dt < best[0]:
            best = (dt, nthr, float(np.abs(flow - table[0]).max()))
    dt, nthr, err = best
    tch = {"value": n / dt, "unit": "triplets/s", "cores": nthr, "kind": "port",
           "sample": "%d triplets at 3x%dx%d, same pruned graph in PyTorch-CPU %s (oneDNN, fp32), best of 16/32/64/128 intra-op "
                     "threads: %d threads, %.2f s; max |flow - oracle| %.1e" % (n, H, W, torch.__version__, nthr, dt, err)}
    return port, tch


def pmc_traffic(B, H, W):
    """HBM bytes per step from the committed PMC passes (profiles/*_traffic.json, produced by
    tools/collect_traffic.sh + tools/traffic_summary.py on this same command); only valid for
    the default workload they were collected on."""
    import glob
    if (B, H, W) != (16, 1024, 1920):
        return {}
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")))
    if not files:
        return {}
    with open(files[-1]) as f:
        d = json.load(f)
    d["file"] = os.path.relpath(files[-1], ROOT)
    return d


def host_path(torch, model, H, W, n=32):
    """PCIe-inclusive rate of the host-buffer boundary (SURVEY 8d "end-to-end incl. H2D/D2H"): computeFlowBatch on n
    triplets in pageable host memory -> f64 flow + u8 masks in host memory.  Reported beside `value`, never as it."""
    import numpy as np
    g = torch.Generator(device="cuda").manual_seed(7)
    by = [torch.randint(0, 256, (n, 3, H, W), generator=g, device="cuda", dtype=torch.uint8).cpu().numpy() for _ in range(3)]
    nf = n // 2
    fl = [torch.rand((nf, 3, H, W), generator=g, device="cuda").cpu().numpy() for _ in range(3)]
    out = (np.empty((n, 2, H, W), np.float64), np.empty((n, 1, H, W), np.uint8), np.empty((n, 1, H, W), np.uint8))

    def best(fn, reps=2):
        fn()                                   # buffers, page faults
        fn()                                   # hipGraph capture (second use of a shape)
        t = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            t.append(time.perf_counter() - t0)
        return min(t)

    t_b = best(lambda: model.computeFlowBatch(*by, out=out))
    o16 = tuple(o[:nf] for o in out)
    t_f = best(lambda: model.computeFlowBatch(*fl, out=o16))
    o1 = tuple(o[:1] for o in out)
    t_1 = best(lambda: model.computeFlowBatch(*[a[:1] for a in by], out=o1), reps=3)
    return {"what": "b2f_compute_flow_batch[_u8]: host buffers in (pageable), f64 flow + u8 masks out, upload / kernels / "
                    "download pipelined; not `value` (that starts from HBM-resident inputs)",
            "unit": "triplets/s", "bytes_in": {"n": n, "value": n / t_b}, "float_in": {"n": nf, "value": nf / t_f},
            "single_triplet_ms_bytes_in": 1e3 * t_1, "finite": bool(np.isfinite(out[0]).all())}


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: N child processes of this same command line, one rank per GPU
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, rendezvous on 127.0.0.1 at a free port).  The parent
    has not initialised the GPU (nothing is exec'd over a process that has); rank 0's stdout is relayed verbatim."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # Poll every child: when one rank dies early (bad GPU, import error) the others would sit in init_process_group / a barrier
    # until the process-group timeout; terminate them instead and report a failure.  Rank 0's stdout goes through a reader thread.
    import threading
    import time
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rcs = [None] * n
    while any(c is None for c in rcs):
        for r, pr in enumerate(procs):
            if rcs[r] is None:
                rcs[r] = pr.poll()
        if any(c not in (None, 0) for c in rcs):
            for r, pr in enumerate(procs):
                if rcs[r] is None:
                    pr.terminate()
            for r, pr in enumerate(procs):
                if rcs[r] is None:
                    try:
                        rcs[r] = pr.wait(timeout=10)
                    except subprocess.TimeoutExpired:
                        pr.kill()
                        rcs[r] = pr.wait()
            break
        time.sleep(0.05)
    reader.join(timeout=10)
    out = b"".join(chunks)
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(rcs) if c != 0]
    if bad:
        sys.stderr.write("bench.py: ranks failed (rank, exit code): %s\n" % bad)
    return 1 if bad else 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=16, help="triplets per GPU per step")
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--graph", type=int, default=1,
                    help="1 (default, BASELINE.json configs[4]): the timed steps replay the captured hipGraph of the forward pass; "
                         "0: eager launches.  Per-kernel times always come from a second, un-timed eager pass with HIP events")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only for tests)")
    ap.add_argument("--share-gpu", action="store_true", help="tests only: every rank uses GPU 0 (needs --backend gloo)")
    ap.add_argument("--no-extras", action="store_true",
                    help="only `value` and the per-kernel profile: no compute_flow_hard_exact, two_pipelines_in_flight, host_path or CPU "
                         "columns -- every forward pass of the process is then the same pass, which is what tools/collect_profiles.sh "
                         "profiles (AverageNs x launches per step of its kernel stats = kernel_ms_per_step)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-path", action="store_true")
    args = ap.parse_args()
    if args.no_extras:
        args.no_cpu_baseline = args.no_host_path = True

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: this process never touches the GPU (torch is not even imported yet); it
        # starts the N ranks as children, relays rank 0's single JSON line and exits with the worst child's code
        sys.exit(spawn_ranks(args.gpus))

    import torch
    import torch.distributed as dist
    from back2future_amd import back2future

    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the computeFlow path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    B, H, W = args.batch, args.height, args.width
    # random-init pwc.lua weights, Ours-Hard shape.  Only rank 0 starts from the benchmark's seed: the other ranks start
    # from different weights, so the broadcast below is load-bearing (a rank it did not reach computes something else
    # and reports another checksum)
    model = back2future.Model("random:hard:%d:1.0" % (2 if rank == 0 else 1000 + rank), device=local_rank)
    bcast = None
    if world > 1:
        # the one collective of the path: RCCL broadcast of the flat weight buffer (28.8 MB) from rank 0,
        # replacing nn.DataParallelTable's NCCL parameter sync (util.lua:27-48)
        from back2future_amd import dist as b2f_dist
        b2f_dist.broadcast_weights(model, src=0)
        mine = torch.tensor([b2f_dist.weights_checksum(model) & 0x7fffffffffffffff], device=dev, dtype=torch.int64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        sums = [int(t.item()) for t in every]
        bcast = {"ranks": world, "checksums_match": len(set(sums)) == 1, "checksum_rank0": "%016x" % sums[0]}

    x = make_triplets(torch, B, H, W, seed=2 + rank, device=dev)
    flow = torch.empty(B, 2, H, W, device=dev)
    occ = torch.empty(B, 2, H, W, device=dev)
    est3 = torch.empty(B, 3, H, W, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    torch.cuda.synchronize()   # x is produced on torch's stream; stream 0 selects the context's own non-blocking stream

    def step():
        model.forward_device(x.data_ptr(), B, H, W, flow.data_ptr(), occ.data_ptr(), est3.data_ptr(),
                             unit_input=True, stream=stream)

    model.set_option("use_graph", args.graph)
    model.set_option("profile", 0)
    for _ in range(max(args.warmup, 3 if args.graph else 0)):   # a graph is captured on the second use of a shape
        step()
    torch.cuda.synchronize()

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    per_rank = None
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        every_t = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every_t, t)
        per_rank = [B * args.steps / float(u.item()) for u in every_t]     # a straggler shows here
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # Beside `value` (never as it): what computeFlow() of an Ours-Hard model really needs -- flow and est[3] = warped frame 1
    # (back2future.lua:77,87 read est[1] and est[3] by position; for a Hard model the occlusion decoder of level 3 feeds
    # only est[2], which computeFlow never looks at).  `value` keeps the occlusion decoder: SURVEY s8d's pruned graph.
    def step_hard_exact():
        model.forward_device(x.data_ptr(), B, H, W, flow.data_ptr(), None, est3.data_ptr(), unit_input=True, stream=stream)
    hard_dt = None
    if not args.no_extras:
        for _ in range(3):
            step_hard_exact()
        torch.cuda.synchronize()
        th0 = time.perf_counter()
        for _ in range(args.steps):
            step_hard_exact()
        torch.cuda.synchronize()
        hard_dt = time.perf_counter() - th0

    # Beside `value` (never as it): two step pipelines in flight -- a second context (own arena, stream and captured graph) takes
    # every other step, so the launch-latency-bound coarse pyramid levels of one step run beside the wide kernels of the next
    two = None
    if world == 1 and args.graph and not args.no_extras:
        try:
            m2 = back2future.Model("random:hard:2:1.0", device=local_rank)
            m2.set_option("use_graph", 1)
            s2 = torch.cuda.Stream()
            x2 = make_triplets(torch, B, H, W, seed=3, device=dev)
            o2 = (torch.empty_like(flow), torch.empty_like(occ), torch.empty_like(est3))
            torch.cuda.synchronize()

            def step2(i):
                if i & 1:
                    m2.forward_device(x2.data_ptr(), B, H, W, o2[0].data_ptr(), o2[1].data_ptr(), o2[2].data_ptr(), unit_input=True, stream=s2.cuda_stream)
                else:
                    step()
            model.set_option("use_graph", 1)
            for i in range(6):
                step2(i)
            torch.cuda.synchronize()
            t20 = time.perf_counter()
            for i in range(args.steps):
                step2(i)
            torch.cuda.synchronize()
            two_dt = time.perf_counter() - t20
            two = {"what": "the same K steps alternating between two contexts / streams on this GPU (independent batches, two captured "
                           "graphs in flight): the coarse levels of one step overlap the wide kernels of the other; reported beside `value`",
                   "value": B * args.steps / two_dt, "unit": "triplets/s", "ms_per_step": 1e3 * two_dt / args.steps}
            m2.close()
            del x2, o2
        except Exception as e:   # never let the extra measurement take the line down
            two = {"error": str(e)}

    # per-kernel times: a second, UN-TIMED pass of the same steps, eager, with HIP events recorded by the library around
    # every launch on the launch stream (the event pairs cost ~2 % and graphs carry no events, so not in `value`)
    model.set_option("use_graph", 0)
    model.set_option("profile", 1)
    step()
    torch.cuda.synchronize()
    model.profile_reset()
    tp0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    prof_dt = time.perf_counter() - tp0
    prof = model.profile_read()
    model.set_option("profile", 0)
    kernels = layer_kernels(model, step, torch)       # the library's kernel class per layer (one more un-timed eager pass)
    finite = bool(torch.isfinite(flow).all().item() and torch.isfinite(occ).all().item())

    if rank == 0:
        px = H * W * B
        out = {
            "metric": "frame-triplets/sec at 3x%dx%d" % (H, W),
            "value": world * B * args.steps / dt,
            "unit": "triplets/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[4] per-GPU share: batch=%d synthetic 3x%dx%d triplets per "
                                   "GPU, random-init pwc.lua weights (Ours-Hard shape), outputs flow + occ + est[3]"
                                   % (B, H, W),
                       "batch_per_gpu": B, "global_batch": B * world, "height": H, "width": W,
                       "parallelism": "dp%d: independent triplets per rank, one RCCL weight broadcast at init" % world,
                       "hip_graph": bool(args.graph)},
            "outputs_finite": finite,
        }
        if bcast:
            out["weights_broadcast"] = bcast
        if per_rank:
            out["per_rank_value"] = {"unit": "triplets/s of each rank over its own wall time (value uses the slowest rank's)", "values": per_rank}
        if hard_dt is not None:
            out["compute_flow_hard_exact"] = {
                "what": "same workload without the level-3 occlusion decoder: everything computeFlow() of an Ours-Hard model reads "
                        "(flow = est[1], masks from est[3] = warped frame 1); reported beside `value`, which keeps SURVEY s8d's pruned "
                        "graph (occlusion decoder of level 3 included)",
                "value": world * B * args.steps / hard_dt if world == 1 else None, "unit": "triplets/s (this rank)" if world > 1 else "triplets/s",
                "rank0_value": B * args.steps / hard_dt, "ms_per_step": 1e3 * hard_dt / args.steps}
        if two:
            out["two_pipelines_in_flight"] = two
        if prof:
            conv_ms = sum(ms for k, (ms, n) in prof.items() if k.startswith("conv")) / args.steps
            conv_n = sum(n for k, (ms, n) in prof.items() if k.startswith("conv")) / args.steps
            corr_ms, corr_n = prof.get("warp_costvol", (0.0, 0))
            corr_ms /= args.steps
            corr_n /= args.steps
            tr = pmc_traffic(B, H, W)
            alg_k, exe_k = conv_flops_by_kernel(H, W, kernels)
            kms = {}
            for k, (ms, n) in prof.items():                      # profile rows -> kernel classes (rows carry an _ntN suffix)
                for cls in alg_k:
                    if k == cls or k.startswith(cls + "_"):
                        kms[cls] = kms.get(cls, 0.0) + ms / args.steps
            dom = max(kms, key=kms.get)                          # dominant kernel class of the step
            d_alg, d_exe, d_ms = alg_k[dom] * B, exe_k[dom] * B, kms[dom]
            a = d_alg / (d_ms * 1e-3) / 1e12
            ea = d_exe / (d_ms * 1e-3) / 1e12
            # `frac` is the utilisation of the matrix pipe: the FLOPs the MFMAs of this kernel really execute (Winograd
            # F(6x6): 64/36, F(4x4): 36/16 MACs per output and channel pair, channel padding included) / its time / the nominal
            # fp32 MFMA peak.  The direct-convolution-equivalent rate (9 MACs) is reported separately as `effective_vs_direct`.
            out["roofline"] = {"kernel": "%s (Winograd %s on the fp32 MFMA; %.0f %% of the profiled step)" % (dom, "F(6x6,3x3)" if dom == "conv3x3_wino6" else "F(4x4,3x3)", 100.0 * d_ms / (1e3 * prof_dt / args.steps))
                               if dom in ("conv3x3_wino4", "conv3x3_wino6") else dom,
                               "bound": "mfma", "achieved": ea, "peak": 157.3, "unit": "TFLOP/s", "frac": ea / 157.3,
                               "peak_note": "nominal dense fp32 matrix peak at 2.4 GHz (MI355X_MICROARCH.md); `frac` is a fraction OF NOMINAL -- "
                                            "under the Winograd kernels the chip holds 2.07 - 2.1 GHz (profiles/r06_wino6_notes.txt)",
                               "executed_flop_per_step": d_exe,
                               "effective_vs_direct": {"achieved": a, "unit": "TFLOP/s of direct-convolution FLOPs (2*9*Ci*Co per output)",
                                                       "x_peak": a / 157.3, "algorithmic_flop_per_step": d_alg},
                               "traffic": tr.get(dom, {}).get("traffic_bytes"), "traffic_unit": "HBM bytes per step (PMC)",
                               "traffic_source": tr.get("file"), "ms_per_step": d_ms,
                               "timing": "HIP events on the launch stream, un-timed eager pass of the same %d steps" % args.steps}
            flops = sum(alg_k.values()) * B
            ex = sum(exe_k.values()) * B
            a_all = flops / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
            e_all = ex / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
            out["roofline_conv_all"] = {"kernel": "all %d conv launches of a step" % conv_n, "bound": "mfma", "achieved": e_all,
                                        "peak": 157.3, "unit": "TFLOP/s", "frac": e_all / 157.3, "executed_flop_per_step": ex,
                                        "effective_vs_direct": {"achieved": a_all, "x_peak": a_all / 157.3, "algorithmic_flop_per_step": flops},
                                        "traffic": tr.get("conv", {}).get("traffic_bytes"), "traffic_unit": "HBM bytes per step (PMC)",
                                        "traffic_source": tr.get("file"), "ms_per_step": conv_ms,
                                        "bf16_pipe_ms_per_step": sum(kms.get(k, 0.0) for k in BF16_PIPE),
                                        "note": "kernels that multiply on the bf16 pipe with split fp32 operands (%s) count with their time and "
                                                "their direct-convolution FLOPs but execute no fp32-MFMA FLOPs" % ", ".join(k for k in BF16_PIPE if k in kms)}
            cb = corr_bytes_per_px() * px
            g = cb / (corr_ms * 1e-3) / 1e9 if corr_ms > 0 else 0.0
            out["roofline_corrwarp"] = {"kernel": "warp_costvol (%d launches of a step)" % corr_n, "bound": "hbm",
                                        "achieved": g, "peak": 8000.0, "unit": "GB/s", "frac": g / 8000.0,
                                        "traffic": tr.get("warp_costvol", {}).get("traffic_bytes"),
                                        "traffic_unit": "HBM bytes per step (PMC)", "traffic_source": tr.get("file"),
                                        "ms_per_step": corr_ms, "algorithmic_bytes_per_step": cb}
            out["kernel_ms_per_step"] = {k: ms / args.steps for k, (ms, n) in sorted(prof.items())}
            out["conv_kernel_of_layer"] = {"what": "kernel class the library ran each conv layer on (cin padded, cout, input map), read back from "
                                                   "its per-layer profile rows -- executed_flop_per_step follows from this, not from a rule restated here",
                                           "layers": {"%dto%d_%dx%d" % k: v for k, v in sorted(kernels.items())}}
            out["profiled_pass_ms_per_step"] = 1e3 * prof_dt / args.steps
        if world == 1 and not args.no_host_path:
            out["host_path"] = host_path(torch, model, H, W)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], out["cpu_baseline_torch"] = cpu_baseline(H, W, 2)
        print(json.dumps(out), flush=True)
    model.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
