"""Build libdexbotic_amd.so (hand-written gfx950 HIP kernels behind the C ABI in include/dexbotic_amd.h).

    python -m dexbotic_amd.build          # incremental, in-tree: dexbotic_amd/libdexbotic_amd.so

hipcc cross-compiles for gfx950 without a GPU.  Objects are cached under dexbotic_amd/csrc/_obj keyed by
source mtime so that re-builds only touch what changed.
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libdexbotic_amd.so")
SOURCES = ["api.cpp", "gemm.hip", "norm.hip", "elementwise.hip", "attention.hip", "optim.hip", "loss.hip", "image.hip", "dit_fused.hip", "memvla.hip", "decode_fused.hip"]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "gemm_common.h"), os.path.join(os.path.dirname(HERE), "include", "dexbotic_amd.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _stale(src: str, obj: str) -> bool:
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(f) > t for f in [src] + HEADERS)


def _compile(name: str) -> str:
    src = os.path.join(CSRC, name)
    obj = os.path.join(OBJ, name + ".o")
    if _stale(src, obj):
        lang = ["-x", "hip"] if name.endswith(".cpp") else []
        cmd = [HIPCC] + FLAGS + lang + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {name}:\n{r.stdout}\n{r.stderr}")
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
    return obj


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    with cf.ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(_compile, SOURCES))
    if force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[dexbotic_amd.build] linked {LIB}")
    elif verbose:
        print(f"[dexbotic_amd.build] up to date: {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
