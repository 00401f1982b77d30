"""GPU: bench.py itself -- the JSON contract of the single-GPU line and the N > 1 code path (two ranks sharing the one
GPU of the test box over gloo: seeds diverge, the weight broadcast lands in the library's device buffer, the checksums
agree, barrier / max-over-ranks timing)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--batch", "2", "--height", "128", "--width", "256", "--steps", "3", "--warmup", "2", "--no-cpu-baseline", "--no-host-path"]


def _line(out):
    lines = [l for l in out.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_single_gpu_line_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL, capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "roofline_corrwarp"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["scaling"] == "weak" and d["dtype"] == "f32" and d["vs_baseline"] is None
    assert d["config"]["hip_graph"] is True and "workload" in d["config"]
    assert 0 < d["roofline"]["frac"] <= 1 and d["roofline"]["bound"] == "mfma"          # a fraction of a roofline, never above 1
    assert 0 < d["roofline_corrwarp"]["frac"] <= 1 and d["roofline_corrwarp"]["bound"] == "hbm"
    assert d["value"] > 0 and d["outputs_finite"]
    # the per-layer kernel classes are read back from the library (not re-derived): every conv layer of the pruned graph is there
    layers = d["conv_kernel_of_layer"]["layers"]
    # (default bf16_direct = 2: the 16 -> 16 and the 16 -> 32 stride-2 layer of the head run as ONE kernel on the bf16 pipe)
    assert layers["16to16_64x128"] == "conv_head16_bf16" and layers["32to2_32x64"] == "conv3x3_narrow2" and layers["16to32_64x128"] == "conv_head16_bf16" and layers["32to64_32x64"] == "conv3x3_s2_bf16"
    assert set(layers.values()) <= {"conv3x3_wino4", "conv3x3_wino", "conv3x3_narrow2", "conv3x3_c16", "conv3x3_s2x16", "conv3x3_s1", "conv3x3_s2",
                                    "conv3x3_c16_bf16", "conv_head16_bf16", "conv3x3_s1_bf16", "conv3x3_s2_bf16", "conv3x3_s2b", "conv3x3_w1b"}
    assert layers["64to96_16x32"] == "conv3x3_s2b"              # stride-2 layers of >= 64 input channels: the loader / consumer kernel (default s2_loader = 1)
    assert "compute_flow_hard_exact" in d and "two_pipelines_in_flight" in d


def test_no_extras_line_is_the_plain_pass_only():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-extras"] + SMALL[:10], capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    for k in ("compute_flow_hard_exact", "two_pipelines_in_flight", "host_path", "cpu_baseline"):
        assert k not in d, k
    assert d["value"] > 0 and "roofline" in d and "kernel_ms_per_step" in d


def test_two_ranks_share_the_gpu_over_gloo():
    port = 29700 + os.getpid() % 200
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpu"] + SMALL
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 4
    assert d["weights_broadcast"]["ranks"] == 2 and d["weights_broadcast"]["checksums_match"] is True
    assert d["value"] > 0 and d["outputs_finite"]


def test_plain_invocation_spawns_ranks():
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE (how the driver's N=1 command line looks with another N):
    the parent starts the ranks itself, relays rank 0's one line and returns their exit code."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpu"] + SMALL,
                       capture_output=True, text=True, cwd=ROOT, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 4
    assert d["weights_broadcast"]["ranks"] == d["n_gpus"] and d["weights_broadcast"]["checksums_match"] is True
    assert len(d["per_rank_value"]["values"]) == 2 and all(v > 0 for v in d["per_rank_value"]["values"])
