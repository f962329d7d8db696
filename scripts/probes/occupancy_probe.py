#!/usr/bin/env python
"""How much of a 256x256 tile's time is contention between the CUs: the ping-pong kernel on grids of 8 .. 1024 tiles at three
depths; per grid the line time = fixed + per_k_tile * (K / 64) through the three depths."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dexbotic_amd import kernels as K  # noqa: E402

def t(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

lay = sys.argv[1] if len(sys.argv) > 1 else "nt"
for tm, tn in [(1, 8), (2, 16), (4, 16), (8, 16), (16, 16), (16, 32), (32, 32), (18, 148)]:
    M, N = 256 * tm, 256 * tn
    row = []
    for Kd in (512, 3584, 7168):
        if lay == "nt":
            a = (torch.rand(M, Kd, device="cuda") * 2 - 1).bfloat16(); b = (torch.rand(N, Kd, device="cuda") * 2 - 1).bfloat16()
            out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); fn = lambda: K.mm_nt(a, b, out=out)
        elif lay == "nn":
            a = (torch.rand(M, Kd, device="cuda") * 2 - 1).bfloat16(); b = (torch.rand(Kd, N, device="cuda") * 2 - 1).bfloat16()
            out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); fn = lambda: K.mm_nn(a, b, out=out)
        else:
            a = (torch.rand(Kd, M, device="cuda") * 2 - 1).bfloat16(); b = (torch.rand(Kd, N, device="cuda") * 2 - 1).bfloat16()
            out = torch.empty(M, N, device="cuda", dtype=torch.float32); fn = lambda: K.mm_tn(a, b, out=out)
        row.append(t(fn))
        del a, b, out
    rounds = -(-tm * tn // 256)
    per = (row[2] - row[1]) / 56 / rounds
    fixed = row[1] - per * 56 * rounds
    print(f"{lay} tiles {tm * tn:5d} ({rounds} rounds): K=512 {row[0]:7.1f}  K=3584 {row[1]:7.1f}  K=7168 {row[2]:7.1f} us -> {per:.3f} us per K tile per round, fixed {fixed:6.1f} us ({fixed / rounds:5.1f} per round); "
          f"K=3584: {2 * M * N * 3584 / row[1] / 1e6:7.1f} TF/s")
