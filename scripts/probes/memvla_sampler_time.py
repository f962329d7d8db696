#!/usr/bin/env python
"""MemVLA's sampler alone at the bench size (DiT-L, perceptual attention over 256 keys, CFG batch 2, 10 DDIM steps): the one-launch kernel
with / without the perceptual-attention phases, timed with events; and whether the model takes it.   python scripts/probes/memvla_sampler_time.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dexbotic_amd import kernels as K  # noqa: E402

DEV = "cuda"


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


def main():
    H, heads, I, depth, steps, P_, N, T, A, nb = 1024, 16, 4096, 24, 10, 256, 2, 16, 7, 1
    ws, ptrs = [], []
    for k in range(depth):
        blk = [rnd(3 * H, H, seed=k, scale=H ** -0.5), rnd(3 * H, seed=k + 100, scale=0.1), rnd(H, H, seed=k + 200, scale=H ** -0.5),
               rnd(H, seed=k + 300, scale=0.1), rnd(I, H, seed=k + 400, scale=H ** -0.5), rnd(I, seed=k + 500, scale=0.1),
               rnd(H, I, seed=k + 600, scale=I ** -0.5), rnd(H, seed=k + 700, scale=0.1), rnd(3 * H, H, seed=k + 800, scale=H ** -0.5),
               rnd(3 * H, seed=k + 900, scale=0.1), rnd(H, H, seed=k + 1000, scale=H ** -0.5), rnd(H, seed=k + 1100, scale=0.1),
               1.0 + rnd(H, seed=k + 1200, scale=0.2), rnd(H, seed=k + 1300, scale=0.2)]
        ws.append(blk)
    x0 = rnd(nb, T, A, seed=7)
    ze, te, pos = rnd(N, H, seed=8, scale=0.5), rnd(steps, H, seed=9, scale=0.5), rnd(T + 1, H, seed=10, scale=0.1)
    xw, xb, fw, fb = rnd(H, A, seed=11, scale=0.3), rnd(H, seed=12, scale=0.1), rnd(A, H, seed=13, scale=H ** -0.5), rnd(A, seed=14, scale=0.1)
    kv = rnd(depth, N, P_, 2, H, seed=15)
    coef = torch.rand(steps, 4, device=DEV) * 0.5 + 0.25
    for per in (False, True):
        table = torch.tensor([w.data_ptr() for blk in ws for w in (blk if per else blk[:8])], dtype=torch.int64).to(DEV)
        arena, ptab = K.dit_bf16_pack(table, depth, H, I, per=per)
        run = lambda: K.dit_sample_bf16_fwd(x0.clone(), ze, te, pos, xw, xb, fw, fb, coef, nb, True, 1.5, ptab, depth, T + 1, H, heads, I, 1e-6,
                                            per_kv=kv if per else None)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        print(f"DiT-L sampler, one launch, perceptual attention {per}: {e0.elapsed_time(e1) / 10:.3f} ms per sample (10 steps, {depth} blocks)"
              f"  dbg={os.environ.get('DXA_DIT_DBG', '0')}", flush=True)
    assert not K.dit_blocks_timed_out()
    if os.environ.get("STAMPS"):          # tuning build (SRC=dit_fused.hip scripts/build_variant.sh ditstamp0 -DDXA_DIT_STAMPS=0; DXA_LIB=_abl/lib_ditstamp0.so)
        import ctypes
        from dexbotic_amd import _lib as L
        lib = ctypes.CDLL(L.LIB_PATH)
        out = (ctypes.c_ulonglong * 48)()
        assert lib.dxa_dit_debug_stamps(out) == 0
        names = ["qkv", "attention", "proj(+out)", "fc1", "fc2", "per-attention"]
        print("cycles per phase (s_memtime, 100 MHz -> x 10 ns) of workgroup 0; product phases: operands+MFMA | partials in LDS | fold+epilogue | ack+assemble | barrier;"
              " per-attention: k/v requested + q in LDS | scores | row softmax | P V | merge + store | barrier")
        for i, n in enumerate(names):
            c = max(out[i * 8 + 7], 1)
            v = [out[i * 8 + j] / c for j in range(6)]
            print(f"  {n:14s} " + " ".join(f"{x:8.0f}" for x in v) + f"   sum {sum(v):8.0f}   ({c} phases)")


if __name__ == "__main__":
    main()
