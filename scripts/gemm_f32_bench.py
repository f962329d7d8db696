#!/usr/bin/env python
"""fp32 (exact v_mfma_f32_16x16x4_f32) GEMM micro-benchmark on the DiT-head shapes of the DB-CogACT step."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dexbotic_amd import kernels as K  # noqa: E402

M = 1088
SHAPES = [("qkv fwd", "nt", M, 2304, 768), ("proj fwd", "nt", M, 768, 768), ("fc1 fwd", "nt", M, 3072, 768),
          ("fc2 fwd", "nt", M, 768, 3072), ("fc1 dX", "nn", M, 768, 3072), ("fc2 dX", "nn", M, 3072, 768),
          ("fc1 dW", "tn", 3072, 768, M), ("fc2 dW", "tn", 768, 3072, M), ("z_emb", "nt", 64, 768, 3584)]


def main():
    for name, lay, m, n, k in SHAPES:
        if lay == "nt":
            a, b, fn = torch.randn(m, k, device="cuda"), torch.randn(n, k, device="cuda"), K.mm_nt
        elif lay == "nn":
            a, b, fn = torch.randn(m, k, device="cuda"), torch.randn(k, n, device="cuda"), K.mm_nn
        else:
            a, b, fn = torch.randn(k, m, device="cuda"), torch.randn(k, n, device="cuda"), K.mm_tn
        out = torch.empty(m, n, device="cuda")
        for _ in range(3):
            fn(a, b, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn(a, b, out=out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"{name:9s} {lay} M={m:5d} N={n:5d} K={k:5d} {ms * 1e3:8.1f} us {2 * m * n * k / ms / 1e9:7.1f} TF/s", flush=True)


if __name__ == "__main__":
    main()
