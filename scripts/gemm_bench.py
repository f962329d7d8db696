#!/usr/bin/env python
"""GEMM micro-benchmark for kernel tuning: times dxa_gemm on the shapes of the DB-CogACT step.

    python scripts/gemm_bench.py                 # default (fast path where eligible)
    DXA_GEMM_NO_FAST=1 python scripts/gemm_bench.py
    DXA_GEMM_FAST_BN=128 python scripts/gemm_bench.py
Random bf16 operands (uniform [-1,1): zero-filled inputs clock ~20 % higher, cdna guide rule 25)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dexbotic_amd import kernels as K  # noqa: E402

M = 4592
SHAPES = [  # (name, layout, M, N, K)
    ("qkv     fwd", "nt", M, 4608, 3584), ("o_proj  fwd", "nt", M, 3584, 3584),
    ("gate_up fwd", "nt", M, 37888, 3584), ("down    fwd", "nt", M, 3584, 18944),
    ("gate_up dX ", "nn", M, 3584, 37888), ("down    dX ", "nn", M, 18944, 3584),
    ("gate_up dW ", "tn", 37888, 3584, M), ("down    dW ", "tn", 3584, 18944, M),
    ("square 4096", "nt", 4096, 4096, 4096), ("square 8192", "nt", 8192, 8192, 8192),
    ("epi K=64   ", "nt", M, 37888, 64), ("epi K=512  ", "nt", M, 37888, 512), ("epi K=1024 ", "nt", M, 37888, 1024),
    ("pre qkv    ", "nt", 543, 4608, 3584), ("pre o_proj ", "nt", 543, 3584, 3584), ("pre gate_up", "nt", 543, 37888, 3584),
    ("pre down   ", "nt", 543, 3584, 18944), ("pre vit fc1", "nt", 514, 4096, 1024), ("pre vit fc2", "nt", 514, 1024, 4096),
    ("pre vit o  ", "nt", 514, 1024, 1024), ("pre vit qkv", "nt", 514, 3072, 1024),
    ("dWgu as nt ", "nt", 37888, 3584, 4608), ("dWdown  nt ", "nt", 3584, 18944, 4608), ("da as nt   ", "nt", M, 18944, 3584),
]


def main():
    dev = "cuda"
    argv = [a for a in sys.argv[1:] if a != "--lib"]
    yard = "--lib" in sys.argv      # yardstick column: torch.matmul (hipBLASLt / rocBLAS) on the same operands — never a product path
    only = argv[0] if argv else None
    for name, lay, m, n, k in SHAPES:
        if only and not any(o == lay or o in name for o in only.split(",")):
            continue
        if lay == "nt":
            a = (torch.rand(m, k, device=dev) * 2 - 1).bfloat16()
            b = (torch.rand(n, k, device=dev) * 2 - 1).bfloat16()
            fn = K.mm_nt
        elif lay == "nn":
            a = (torch.rand(m, k, device=dev) * 2 - 1).bfloat16()
            b = (torch.rand(k, n, device=dev) * 2 - 1).bfloat16()
            fn = K.mm_nn
        else:
            a = (torch.rand(k, m, device=dev) * 2 - 1).bfloat16()
            b = (torch.rand(k, n, device=dev) * 2 - 1).bfloat16()
            fn = K.mm_tn
        out = torch.empty(m, n, device=dev, dtype=torch.float32 if lay == "tn" else torch.bfloat16)   # dW goes to the fp32 arena
        for _ in range(3):
            fn(a, b, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record()
        for _ in range(reps):
            fn(a, b, out=out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        extra = ""
        if yard:
            ta, tb = (a.t() if lay == "tn" else a), (b.t() if lay == "nt" else b)
            for _ in range(3):
                torch.matmul(ta, tb, out=out)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(reps):
                torch.matmul(ta, tb, out=out)
            e1.record()
            torch.cuda.synchronize()
            lms = e0.elapsed_time(e1) / reps
            extra = f"   | library {lms*1e3:9.1f} us {2*m*n*k/lms/1e9:8.1f} TFLOP/s"
        print(f"{name} {lay} M={m:6d} N={n:6d} K={k:6d}  {ms*1e3:9.1f} us  {2*m*n*k/ms/1e9:8.1f} TFLOP/s{extra}", flush=True)


if __name__ == "__main__":
    main()
