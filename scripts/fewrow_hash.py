#!/usr/bin/env python
"""CRC of few-row NT products (a one-request prefill: 543 / 287 / 514 rows against the decoder's and the ViT's matrices, with bias,
residual, fp32 / bf16 outputs, repeated launches: the split-K arrival counters must come back to zero): run it under two builds or
switches (DXA_GEMM_NO_T128, DXA_GEMM_NO_SPLIT, DXA_SPLIT_MAX ...) and diff the lines — a change of the split-K hand-off must leave
every bit where it was.  Also checks each result against an fp64 product (max error / max |ref|)."""
import os
import sys
import zlib

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dexbotic_amd import kernels as K  # noqa: E402

CASES = [(543, 4608, 3584, "bias"), (543, 3584, 3584, "res"), (543, 3584, 18944, "res"), (287, 4608, 3584, "bias"), (287, 3584, 18944, "res"),
         (287, 3584, 3584, "plain"), (543, 1024, 4096, "both"), (514, 1024, 1024, "plain"), (200, 520, 1024, "both"), (543, 768, 2304, "plain"),
         (1088, 768, 9216, "bias")]


def crc(t):
    return zlib.crc32(t.detach().cpu().contiguous().view(torch.uint8).numpy().tobytes())


for m, n, k, epi in CASES:
    g = torch.Generator(device="cuda").manual_seed(m * 7 + n * 3 + k)
    r = lambda *s: (torch.rand(*s, device="cuda", generator=g) * 2 - 1).bfloat16()
    a, b = r(m, k), r(n, k)
    bias = r(n) if epi in ("bias", "both") else None
    res = r(m, n) if epi in ("res", "both") else None
    ref = a.double() @ b.double().t()
    if bias is not None:
        ref += bias.double()
    if res is not None:
        ref += res.double()
    for odt in (torch.bfloat16, torch.float32):
        if odt == torch.float32 and res is not None:
            continue                                   # (residual dtype = input dtype on the bf16 menu; the fp32-out case is dW-style)
        crcs = set()
        for rep in range(4):                           # the counters must come back to zero: four launches, one answer
            out = torch.full((m, n), float("nan"), device="cuda", dtype=odt)
            K.mm_nt(a, b, out=out, bias=bias, residual=res)
            crcs.add(crc(out))
        err = ((out.double() - ref).abs().max() / ref.abs().max()).item()
        print(f"nt {m} {n} {k} {epi} {str(odt)[6:]}: {'/'.join(f'{c:08x}' for c in sorted(crcs))} err {err:.2e}", flush=True)
