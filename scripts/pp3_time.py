#!/usr/bin/env python
"""main-loop rate of the 192-row kernels: time of a 543 x 37888 product at K = 3584 and 7168 (the difference is 56 K tiles of main loop
in each of the two rounds of tiles), graph-replayed over 4 weight sets"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dexbotic_amd import kernels as K  # noqa: E402
dev = "cuda"
for M in [int(v) for v in os.environ.get("ROWS", "543").split(",")]:
    res = {}
    for Kd in (3584, 7168):
        N = 37888
        ws = [(torch.randn(N, Kd, device=dev) * 0.02).bfloat16() for _ in range(4)]
        x = torch.randn(M, Kd, device=dev).bfloat16()
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
            gr.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(10):
                gr.replay()
            e1.record(s)
        torch.cuda.synchronize()
        res[Kd] = 1e3 * e0.elapsed_time(e1) / 40
        del ws
    per_ktile = (res[7168] - res[3584]) / 2 / 56
    print(f"M={M}: K=3584 {res[3584]:.1f} us, K=7168 {res[7168]:.1f} us -> {per_ktile:.3f} us per K tile of 64 per round, fixed part {res[3584] - 2 * 56 * per_ktile:.1f} us")
