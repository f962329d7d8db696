#!/usr/bin/env python
"""AdamW / sumsq micro-benchmark (HBM-bound kernels): reports achieved TB/s on a 2 G-element arena.

    python scripts/adamw_bench.py [chunk_elems]
adamw moves 30 B/element (p,g,m,v read; p,m,v written; bf16 shadow written), sumsq 4 B/element."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dexbotic_amd import kernels as K  # noqa: E402


def main():
    n = 2 * 1024 ** 3
    chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
    dev = "cuda"
    p = torch.randn(n, device=dev)
    g = torch.randn(n, device=dev) * 1e-3
    m = torch.zeros(n, device=dev)
    v = torch.zeros(n, device=dev)
    sh = torch.empty(n, device=dev, dtype=torch.bfloat16)
    starts = torch.arange(0, n, chunk, dtype=torch.int64, device=dev)
    lens = torch.full((starts.numel(),), chunk, dtype=torch.int32, device=dev)
    grp = torch.zeros(starts.numel(), dtype=torch.int32, device=dev)
    clip = torch.ones(1, device=dev)
    out = torch.zeros(1, device=dev)
    scratch = torch.zeros(4096, device=dev, dtype=torch.float64)

    def run_adam(step):
        K.adamw(p, g, m, v, sh, starts, lens, grp, [1e-5], [0.01], 0.9, 0.999, 1e-8, step, clip=clip)

    for name, fn, bpe in (("adamw", run_adam, 30), ("sumsq", lambda s: K.sumsq(g, out, scratch), 4)):
        for s in range(1, 3):
            fn(s)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        e0.record()
        for s in range(reps):
            fn(3 + s)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print(f"{name}: {ms:8.3f} ms  {n * bpe / ms / 1e9:6.2f} TB/s  (chunk {chunk})", flush=True)


if __name__ == "__main__":
    main()
