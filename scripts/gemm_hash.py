#!/usr/bin/env python
"""CRC of the outputs of a fixed list of 16-bit products (the step's shapes, ragged / split cases, the dW epilogue menu, the
two-segment TN product): run it with two builds of the library (DXA_LIB=...) and diff the lines — a schedule change of the
ping-pong kernel must leave every bit where it was."""
import os
import sys
import zlib

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dexbotic_amd import kernels as K  # noqa: E402

M = 4592
CASES = [("nt", M, 4608, 3584), ("nt", M, 3584, 3584), ("nt", M, 37888, 3584), ("nt", M, 3584, 18944), ("nn", M, 3584, 4608),
         ("nn", M, 3584, 37888), ("nn", M, 18944, 3584), ("tn", 4608, 3584, M), ("tn", 37888, 3584, M), ("tn", 3584, 18944, M),
         ("nt", 2296, 3584, 3584), ("nn", 2296, 3584, 18944), ("tn", 4608, 3584, 2296), ("nt", 300, 520, 192), ("nn", 513, 264, 128),
         ("tn", 777, 1032, 4592), ("nt", 256, 256, 64), ("tn", 1152, 4304, 4112), ("nt", 4112, 1152, 4288), ("nn", 4112, 1024, 4096)]


def crc(t):
    return zlib.crc32(t.detach().cpu().contiguous().view(torch.uint8).numpy().tobytes())


def ops(lay, m, n, k):
    g = torch.Generator(device="cuda").manual_seed(m * 7 + n * 3 + k)
    r = lambda *s: (torch.rand(*s, device="cuda", generator=g) * 2 - 1).bfloat16()
    return (r(m, k), r(n, k), K.mm_nt) if lay == "nt" else ((r(m, k), r(k, n), K.mm_nn) if lay == "nn" else (r(k, m), r(k, n), K.mm_tn))


for lay, m, n, k in CASES:
    a, b, fn = ops(lay, m, n, k)
    for odt in (torch.bfloat16, torch.float32):
        out = torch.empty(m, n, device="cuda", dtype=odt)
        fn(a, b, out=out)
        line = f"{lay} {m} {n} {k} {str(odt)[6:]}: {crc(out):08x}"
        if lay == "tn" and odt == torch.float32:          # the dW epilogue menu: accumulate + bf16 mirror + sum-of-squares partials
            c = torch.rand(m, n, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
            mir = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
            ssq = torch.zeros(K.L.lib.dxa_gemm_sumsq_slots(m, n), device="cuda")
            fn(a, b, out=c, accumulate=True, mirror=mir, sumsq=ssq)
            line += f"  acc {crc(c):08x} mirror {crc(mir):08x} sumsq {crc(ssq):08x}"
            k2 = 2296
            g = torch.Generator(device="cuda").manual_seed(11)
            a2 = (torch.rand(k2, m, device="cuda", generator=g) * 2 - 1).bfloat16()
            b2 = (torch.rand(k2, n, device="cuda", generator=g) * 2 - 1).bfloat16()
            o2 = torch.empty(m, n, device="cuda")
            fn(a, b, out=o2, a2=a2, b2=b2)
            line += f" two-segment {crc(o2):08x}"
        print(line, flush=True)
