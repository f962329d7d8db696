#!/usr/bin/env python
"""per-shape time of the few-row fp32 linears of a DiT-L sampler step (34 rows x {qkv, proj, fc1, fc2}) over 24 different
weight sets (nothing stays cached between uses, like the 24 blocks of the head):  python scripts/skinny_bench.py
(run under rocprofv3 --kernel-trace for the per-kernel table; prints the event-timed mean per shape too)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dexbotic_amd import kernels as K  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    M = int(os.environ.get("ROWS", "34"))
    shapes = [("qkv", 3072, 1024), ("proj", 1024, 1024), ("fc1", 4096, 1024), ("fc2", 1024, 4096)]
    g = torch.Generator(device="cpu").manual_seed(0)
    for name, N, Kd in shapes:
        ws = [torch.randn(N, Kd, generator=g).to(dev) * 0.02 for _ in range(24)]
        b = torch.zeros(N, device=dev)
        x = torch.randn(M, Kd, generator=g).to(dev)
        out = torch.empty(M, N, device=dev)
        gr = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for w in ws[:4]:                       # (the stream's split-K scratch comes into being outside the capture)
                K.mm_nt(x, w, bias=b, out=out)
            torch.cuda.synchronize()
            with torch.cuda.graph(gr, stream=s):
                for w in ws:
                    K.mm_nt(x, w, bias=b, out=out)
            gr.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(10):
                gr.replay()
            e1.record(s)
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / 240
        print(f"{name:5s} M={M} N={N} K={Kd}: {us:7.2f} us/call  ({N * Kd * 4 / us / 1e6:6.2f} TB/s of weights)", flush=True)


if __name__ == "__main__":
    main()
