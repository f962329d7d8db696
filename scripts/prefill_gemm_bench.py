#!/usr/bin/env python
"""per-shape time of the decoder's four linears at the row count of a one-request prefill (Qwen2.5-7B widths), each over 8
different weight sets (nothing stays cached between uses):  ROWS=287 python scripts/prefill_gemm_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dexbotic_amd import kernels as K  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    shapes = [("qkv", 4608, 3584), ("o", 3584, 3584), ("gate_up", 37888, 3584), ("down", 3584, 18944)]
    if os.environ.get("SHAPES") == "vit":          # CLIP-L/14 on two views: 514 rows (ROWS=514)
        shapes = [("v_qkv", 3072, 1024), ("v_proj", 1024, 1024), ("v_fc1", 4096, 1024), ("v_fc2", 1024, 4096)]
    for M in [int(v) for v in os.environ.get("ROWS", "287").split(",")]:
        for name, N, Kd in shapes:
            ws = [(torch.randn(N, Kd, device=dev) * 0.02).to(torch.bfloat16) for _ in range(8 if N * Kd > (1 << 24) else 24)]
            x = torch.randn(M, Kd, device=dev).to(torch.bfloat16)
            out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for w in ws[:2]:
                    K.mm_nt(x, w, out=out)
                torch.cuda.synchronize()
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, stream=s):
                    for w in ws:
                        K.mm_nt(x, w, out=out)
                gr.replay()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(s)
                for _ in range(10):
                    gr.replay()
                e1.record(s)
            torch.cuda.synchronize()
            us = 1e3 * e0.elapsed_time(e1) / (10 * len(ws))
            print(f"M={M:4d} {name:8s} N={N:6d} K={Kd:6d}: {us:8.2f} us  {N * Kd * 2 / us / 1e6:5.2f} TB/s weights  "
                  f"{2.0 * M * N * Kd / us / 1e6:7.1f} TF/s useful", flush=True)
            del ws


if __name__ == "__main__":
    main()
