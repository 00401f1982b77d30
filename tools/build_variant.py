#!/usr/bin/env python3
"""Profiling variant of libb2f.so: recompiles ONE source with extra flags and links it with the
regular objects.   python tools/build_variant.py <name> <source.hip> <flag> [<flag> ...]
-> back2future_amd/libb2f_<name>.so  (select with B2F_LIB=<path>; results of ablated builds are wrong).
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from back2future_amd import build as B  # noqa: E402


def main():
    name, src, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
    B.build()
    obj = os.path.join(B.BUILD, "%s_%s.o" % (os.path.splitext(src)[0], name))
    subprocess.check_call([B.HIPCC] + B.FLAGS + B.EXTRA.get(src, []) + flags + ["-x", "hip", "-c", os.path.join(B.CSRC, src), "-o", obj])
    objs = [obj if s == src else os.path.join(B.BUILD, os.path.splitext(s)[0] + ".o") for s in B.SOURCES]
    out = os.path.join(B.HERE, "libb2f_%s.so" % name)
    subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + ["-ldl", "-lpthread"])
    print(out)


if __name__ == "__main__":
    main()
