#!/usr/bin/env python
"""CRC + time of the head-sized fp32 attention backward at MemVLA's perceptual shape (16 samples x 16 heads, 68 queries, 256 keys, D 64)
and two ragged shapes: run with and without DXA_ATTN_SMALL_BWD_V1=1 and diff — the register-blocked kernel sums in the same order."""
import os, sys, zlib, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dexbotic_amd import kernels as K  # noqa: E402
crc = lambda t: zlib.crc32(t.detach().cpu().contiguous().view(torch.uint8).numpy().tobytes())
for B, H, Sq, Sk, D in [(16, 16, 68, 256, 64), (3, 5, 96, 70, 64), (2, 4, 33, 200, 32)]:
    g = torch.Generator(device="cuda").manual_seed(Sq * 7 + Sk)
    mk = lambda S: torch.randn(B, S, H, D, device="cuda", generator=g).permute(0, 2, 1, 3)
    q, k, v, do = mk(Sq), mk(Sk), mk(Sk), mk(Sq)
    o = torch.empty(B, Sq, H, D, device="cuda").permute(0, 2, 1, 3)
    lse = K.attn_fwd(q, k, v, o, causal=False, scale=D ** -0.5)
    dq, dk, dv = (torch.empty(B, S, H, D, device="cuda").permute(0, 2, 1, 3) for S in (Sq, Sk, Sk))
    for _ in range(3):
        K.attn_bwd(q, k, v, o, lse, do, dq, dk, dv, causal=False, scale=D ** -0.5)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        K.attn_bwd(q, k, v, o, lse, do, dq, dk, dv, causal=False, scale=D ** -0.5)
    e1.record(); torch.cuda.synchronize()
    print(f"B {B} H {H} Sq {Sq} Sk {Sk} D {D}: dq {crc(dq):08x} dk {crc(dk):08x} dv {crc(dv):08x}   {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call")
