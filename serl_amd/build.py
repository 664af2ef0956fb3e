"""Builds libserl_mi355.so (HIP, gfx950 only) in-tree with hipcc.  No JIT cache: the .so
travels with the repo snapshot to the GPU box."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libserl_mi355.so")
SOURCES = ["replay.hip", "trunk.hip", "trunk_f16x3.hip", "heads.hip", "small_encoder.hip", "agent.hip", "classifier.hip", "prof.hip", "jaxrng.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _stale(obj, srcs):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(s) > t for s in srcs)


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIB_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(HERE, "..", "include", "serl_mi355.h"))
    objs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        if not os.path.exists(sp):
            continue
        obj = os.path.join(LIB_DIR, src.replace(".hip", ".o"))
        if force or _stale(obj, [sp] + headers):
            cmd = [hipcc] + [f for f in FLAGS if f] + ["-c", sp, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        objs.append(obj)
    if force or _stale(LIB_PATH, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
