#!/usr/bin/env python
"""What would a transposed operand buy the dW product?  dW = dY^T X on the decoder's shapes as it runs today (TN: both operands k-strided,
fp32 out) against the same contraction with dY^T given (NN: A k-contiguous, B = X k-strided, fp32 out) and with both transposed (NT).
8 operand sets rotated (no L2 / Infinity-Cache residency between launches), 40 launches each, random bf16 operands."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dexbotic_amd import kernels as K  # noqa: E402

T = 4608        # tokens padded to a K tile (4592 -> 4608) so that every layout is admissible
SHAPES = [("down    dW", 3584, 18944), ("gate_up dW", 37888, 3584), ("qkv     dW", 4608, 3584), ("o_proj  dW", 3584, 3584)]


def bench(fn, sets, out):
    for s in sets[:2]:
        fn(*s, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 40
    e0.record()
    for i in range(reps):
        fn(*sets[i % len(sets)], out=out)
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


def main():
    dev = "cuda"
    nset = 4
    for name, m, n in SHAPES:
        out = torch.empty(m, n, device=dev, dtype=torch.float32)
        dy = [(torch.rand(T, m, device=dev) * 2 - 1).bfloat16() for _ in range(nset)]
        x = [(torch.rand(T, n, device=dev) * 2 - 1).bfloat16() for _ in range(nset)]
        dyt = [d.t().contiguous() for d in dy]
        xt = [v.t().contiguous() for v in x]
        t_tn = bench(K.mm_tn, list(zip(dy, x)), out)
        t_nn = bench(K.mm_nn, list(zip(dyt, x)), out)
        t_nt = bench(K.mm_nt, list(zip(dyt, xt)), out)
        fl = 2.0 * m * n * T
        print(f"{name} M={m:6d} N={n:6d} K={T}:  TN {t_tn:8.1f} us {fl / t_tn / 1e6:7.1f} TF/s | dY^T given (NN) {t_nn:8.1f} us {fl / t_nn / 1e6:7.1f} | "
              f"both (NT) {t_nt:8.1f} us {fl / t_nt / 1e6:7.1f}", flush=True)
        del dy, x, dyt, xt, out


if __name__ == "__main__":
    main()
