#!/usr/bin/env python
"""Yardstick only (never on the product path): what torch.matmul (hipBLASLt / rocBLAS) reaches on the step's GEMM shapes,
to price dexbotic_amd's ring kernel against the vendor library on the same box, clocks and operand values."""
import sys

import torch

SHAPES = [("qkv", 4592, 4608, 3584), ("o_proj", 4592, 3584, 3584), ("gate_up", 4592, 37888, 3584), ("down", 4592, 3584, 18944),
          ("sq8192", 8192, 8192, 8192)]


def main():
    dev = "cuda"
    for name, m, n, k in SHAPES:
        a = (torch.rand(m, k, device=dev) * 2 - 1).bfloat16()
        b = (torch.rand(n, k, device=dev) * 2 - 1).bfloat16()
        for _ in range(3):
            torch.matmul(a, b.t())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            torch.matmul(a, b.t())
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"{name:8s} M={m} N={n} K={k}: {ms*1e3:8.1f} us {2*m*n*k/ms/1e9:8.1f} TFLOP/s (torch.matmul NT)", flush=True)


if __name__ == "__main__":
    sys.exit(main())
