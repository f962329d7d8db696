#!/usr/bin/env python
"""Attention forward/backward micro-benchmark on the DB-CogACT step's shapes (bf16)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dexbotic_amd import kernels as K  # noqa: E402

CASES = [("qwen2 causal gqa", 16, 28, 4, 287, 128, True), ("clip vit", 16, 16, 16, 257, 64, False),
         ("prefill B=1", 1, 28, 4, 543, 128, True), ("siglip hd72", 48, 16, 16, 256, 72, False), ("pi0 hd256", 16, 8, 1, 816, 256, False)]


def timeit(fn, reps=20):
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


def sums(*ts):
    """order-independent fingerprints of the outputs (float64 sum and sum of squares): the variants of one kernel accumulate every
    element in the same order, so their fingerprints must be EQUAL, not merely close"""
    return " ".join(f"{t.double().sum().item():.10e}/{(t.double() ** 2).sum().item():.10e}" for t in ts)


def main():
    torch.manual_seed(0)
    for name, B, Hq, Hkv, S, D, causal in CASES:
        mk = lambda h: (torch.randn(B, h, S, D, device="cuda") * 0.5).bfloat16()
        q, k, v, do = mk(Hq), mk(Hkv), mk(Hkv), mk(Hq)
        o = torch.empty_like(q)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        scale = D ** -0.5
        lse = K.attn_fwd(q, k, v, o, causal=causal, scale=scale)
        flops = 4.0 * B * Hq * S * S * D * (0.5 if causal else 1.0)
        t_f = timeit(lambda: K.attn_fwd(q, k, v, o, causal=causal, scale=scale))
        t_b = timeit(lambda: K.attn_bwd(q, k, v, o, lse, do, dq, dk, dv, causal=causal, scale=scale))
        print(f"{name:18s} fwd {t_f:7.1f} us {flops / t_f / 1e6:6.1f} TF/s | bwd {t_b:7.1f} us {2.5 * flops / t_b / 1e6:6.1f} TF/s", flush=True)
        if os.environ.get("ATTN_BENCH_SUMS"):
            print(f"   sums o, dq, dk, dv: {sums(o, dq, dk, dv)}", flush=True)
        if "pi0" in name:
            # the masks of the pi0 training step: block-prefix limits (prefix 784 | state | actions) and per-key validity
            lim = torch.full((B, S), S - 17, dtype=torch.int32, device="cuda")
            lim[:, S - 17] = S - 16
            lim[:, S - 16:] = S
            valid = torch.ones(B, S, dtype=torch.uint8, device="cuda")
            valid[1, 256:512] = 0
            kw = dict(causal=False, scale=scale, q_limit=lim, key_valid=valid)
            lse = K.attn_fwd(q, k, v, o, **kw)
            t_f = timeit(lambda: K.attn_fwd(q, k, v, o, **kw))
            t_b = timeit(lambda: K.attn_bwd(q, k, v, o, lse, do, dq, dk, dv, **kw))
            print(f"{name + ' masked':18s} fwd {t_f:7.1f} us {flops / t_f / 1e6:6.1f} TF/s | bwd {t_b:7.1f} us "
                  f"{2.5 * flops / t_b / 1e6:6.1f} TF/s", flush=True)
            if os.environ.get("ATTN_BENCH_SUMS"):
                print(f"   sums o, dq, dk, dv: {sums(o, dq, dk, dv)}", flush=True)


if __name__ == "__main__":
    main()
