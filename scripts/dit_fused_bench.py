#!/usr/bin/env python
"""Times the persistent DiT-block kernel at DiT-B size (12 blocks, 2 x 17 rows) against the block-by-block kernels.
    python scripts/dit_fused_bench.py           (DXA_DIT_GRID=n caps the grid of the fused kernel)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dexbotic_amd import kernels as K  # noqa: E402


def main():
    N, T1, H, heads, I, depth = int(os.environ.get("DIT_N", "2")), int(os.environ.get("DIT_T", "17")), 768, 12, 3072, 12
    dev = "cuda"
    g = torch.Generator().manual_seed(0)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
    ws, ptrs = [], []
    for _ in range(depth):
        blk = [r(3 * H, H, sc=H ** -0.5), r(3 * H, sc=0.1), r(H, H, sc=H ** -0.5), r(H, sc=0.1), r(I, H, sc=H ** -0.5),
               r(I, sc=0.1), r(H, I, sc=I ** -0.5), r(H, sc=0.1)]
        ws.append(blk)
        ptrs += [w.data_ptr() for w in blk]
    table = torch.tensor(ptrs, dtype=torch.int64).to(dev)
    h0 = r(N * T1, H)

    def fused():
        return K.dit_blocks_fwd(h0.clone(), table, depth, N, T1, H, heads, I, 1e-6)

    def unfused():
        h = h0.clone()
        for qw, qb, pw, pb, w1, b1, w2, b2 in ws:
            y, _, _ = K.layernorm_fwd(h, None, None, 1e-6)
            qkv = K.mm_nt(y, qw, bias=qb).view(N, T1, 3, heads, 64)
            q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
            o = torch.empty((N, T1, heads, 64), device=dev)
            K.attn_fwd(q, k, v, o.permute(0, 2, 1, 3), causal=False, scale=0.125)
            h = K.mm_nt(o.view(N * T1, H), pw, bias=pb, residual=h)
            y, _, _ = K.layernorm_fwd(h, None, None, 1e-6)
            a = K.mm_nt(y, w1, bias=b1, act=2)
            h = K.mm_nt(a, w2, bias=b2, residual=h)
        return h

    a, b = fused(), unfused()
    print("max abs diff fused vs unfused:", float((a - b).abs().max()), "scale", float(b.abs().max()))
    for name, fn in (("fused", fused), ("unfused", unfused)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(f"{name}: {e0.elapsed_time(e1) / 30 * 1e3:.1f} us per 12-block forward", flush=True)


if __name__ == "__main__":
    main()
