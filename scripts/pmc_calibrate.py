#!/usr/bin/env python
"""Known-byte-count launches of the ping-pong GEMM for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on ITS access pattern
(LDS-DMA of 16 B per lane, 8 lanes per 128-byte line in permuted chunk order; 16-byte buffer stores):

    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_cal -o fetch -- python scripts/pmc_calibrate.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc_cal -o write -- python scripts/pmc_calibrate.py

case "stream": C[256, 65536] = A[256, 4096] B[65536, 4096]^T — one row of tiles: every byte of B (512 MiB, twice the 256 MiB
Infinity Cache) is needed by exactly one workgroup, A (2 MiB) lives in L2: the HBM read traffic is B once (+ A once).
case "store":  C[4096, 16384] = A[4096, 64] B[16384, 64]^T — 128 MiB of bf16 output written exactly once, 2.5 MiB read.
The expected byte counts are printed; the ratio reported / expected is the correction `bench.py` applies to the counters of
the same kernel in the training step (profiles/r02_pmc.json)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dexbotic_amd import kernels as K  # noqa: E402

CASES = {"stream": (256, 65536, 4096), "store": (4096, 16384, 64)}


def main():
    dev = "cuda"
    for name, (m, n, k) in CASES.items():
        a = (torch.rand(m, k, device=dev) * 2 - 1).bfloat16()
        b = (torch.rand(n, k, device=dev) * 2 - 1).bfloat16()
        out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
        for _ in range(3):
            K.mm_nt(a, b, out=out)
        torch.cuda.synchronize()
        print(f"{name}: M={m} N={n} K={k}  expected read {(m * k + n * k) * 2} B, expected write {m * n * 2} B per launch", flush=True)


if __name__ == "__main__":
    main()
