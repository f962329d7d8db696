#!/usr/bin/env python
"""CRC of bf16 NT products whose row count picks 192-row tiles (dxa_gemm: pad192 * 27 < pad256 * 25): run once with the ping-pong
192-row kernel (default) and once with DXA_GEMM_PP3=0 (ring kernel) and diff — same MFMA, same K order, same epilogue: every bit
must agree; each line also carries the distance to an fp32 torch product."""
import sys, os, zlib
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dexbotic_amd import kernels as K  # noqa: E402

CASES = [(543, 37888, 3584), (543, 3584, 18944), (543, 4608, 3584), (543, 3584, 3584), (576, 37888, 3584), (300, 520, 192), (192, 256, 64),
         (190, 264, 128), (543, 1000, 3584), (1100, 3584, 1024), (2300, 4608, 3584), (543, 3584, 64), (543, 3584, 128), (543, 3584, 320)]


def crc(t):
    return zlib.crc32(t.detach().cpu().contiguous().view(torch.uint8).numpy().tobytes())


for m, n, k in CASES:
    g = torch.Generator(device="cuda").manual_seed(m * 7 + n * 3 + k)
    r = lambda *s: (torch.rand(*s, device="cuda", generator=g) * 2 - 1).bfloat16()
    a, b, bias, res = r(m, k), r(n, k), r(n), r(m, n)
    ref = a.float() @ b.float().t()
    line = f"{m} {n} {k}:"
    for odt in (torch.bfloat16, torch.float32):
        out = torch.empty(m, n, device="cuda", dtype=odt)
        K.mm_nt(a, b, out=out)
        err = float((out.float() - ref).abs().max() / ref.abs().max())
        out2 = torch.empty(m, n, device="cuda", dtype=odt)
        K.mm_nt(a, b, out=out2, bias=bias, residual=res if odt == torch.bfloat16 else None)
        line += f" {str(odt)[6:]} {crc(out):08x} err {err:.1e} epi {crc(out2):08x}"
    print(line, flush=True)
