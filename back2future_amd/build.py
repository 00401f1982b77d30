"""Builds back2future_amd/libb2f.so (HIP/gfx950 kernels + C ABI) in-tree with hipcc.

    python -m back2future_amd.build [--force]

hipcc cross-compiles for gfx950 without a GPU.  The .so is git-ignored but travels to
the GPU box with the gpurun snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libb2f.so")
BUILD = os.path.join(HERE, "build")
SOURCES = ["b2f_conv.hip", "b2f_wino.hip", "b2f_wino4.hip", "b2f_wino6.hip", "b2f_w1b.hip", "b2f_s2b.hip", "b2f_conv16.hip", "b2f_head.hip", "b2f_convb.hip", "b2f_corr.hip", "b2f_corr5.hip", "b2f_glue.hip", "b2f_boundary.hip", "b2f_api.hip", "b2f_graph.hip", "b2f_backward.hip", "b2f_pipeline.hip", "b2f_multi.hip", "b2f_host.cpp", "b2f_t7.cpp"]
HEADERS = ["b2f_internal.h", "b2f_host.h", "b2f_ctx.h", "b2f_corr5_loop.inc", os.path.join("..", "..", "include", "b2f.h")]
# tools/experiments/csrc: kernels that were built, tested and measured no faster than the defaults; `--experiments` builds them and
# the options that select them into libb2f_exp.so (B2F_LIB=<that file>); the product library does not contain them
EXP_DIR = os.path.join(os.path.dirname(HERE), "tools", "experiments", "csrc")
EXP_SOURCES = ["b2f_wino4s.hip", "b2f_wino2s.hip", "b2f_conv16b.hip"]
EXP_OUT = os.path.join(HERE, "libb2f_exp.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fvisibility=hidden", "-Wall", "-Wno-unused-result"]
# b2f_corr: the SLP vectorizer packs the 81 independent FMA chains into v_pk_fma_f32 pairs whose
# operands are not register-adjacent, which costs ~2 v_mov per FMA; plain v_fmac is faster here.
# b2f_wino4s: packed fp32 ops do not overlap the bf16 MFMAs (tools/mfma_bf16_chain.hip); its VALU work is written scalar on purpose.
EXTRA = {"b2f_corr.hip": ["-fno-slp-vectorize"], "b2f_corr5.hip": ["-fno-slp-vectorize"], "b2f_wino4s.hip": ["-fno-slp-vectorize"], "b2f_wino2s.hip": ["-fno-slp-vectorize"], "b2f_conv16b.hip": ["-fno-slp-vectorize"], "b2f_head.hip": ["-fno-slp-vectorize"], "b2f_convb.hip": ["-fno-slp-vectorize"], "b2f_w1b.hip": ["-fno-slp-vectorize"], "b2f_s2b.hip": ["-fno-slp-vectorize"]}



def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, experiments=False):
    bdir = BUILD + ("_exp" if experiments else "")
    target = EXP_OUT if experiments else OUT
    os.makedirs(bdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    procs = []
    srcs = [(s, os.path.join(CSRC, s)) for s in SOURCES] + ([(s, os.path.join(EXP_DIR, s)) for s in EXP_SOURCES] if experiments else [])
    for src, sp in srcs:
        obj = os.path.join(bdir, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        if force or _stale(obj, [sp] + hdrs):
            cmd = [HIPCC] + FLAGS + (["-DB2F_EXPERIMENTS=1", "-I", CSRC] if experiments else []) + EXTRA.get(src, []) + (["-x", "hip"] if src.endswith(".hip") else []) + ["-c", sp, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write("hipcc failed on %s:\n%s\n" % (src, out.decode()))
        elif verbose and out.strip():
            print(out.decode())
    if failed:
        raise RuntimeError("libb2f build failed")
    if force or procs or _stale(target, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", target] + objs + ["-ldl", "-lpthread"]
        subprocess.check_call(cmd)
    return target


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, experiments="--experiments" in sys.argv))
