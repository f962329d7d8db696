#!/usr/bin/env python
"""Reads the barrier-to-barrier intervals the stamp build of the ping-pong kernel leaves in C (scripts/ablate_gemm.sh
stamp:"-DDXA_PPV=128"; run with DXA_LIB=_abl/lib_stamp.so).  Interval i of a wave = cycles between its barrier exits i and i+1
within a K tile (0: C0, 1: M1, 2: C1, 3: M2, 4: C2, 5: M3, 6: C3, 7: M0 of the next tile), averaged over the K tiles."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dexbotic_amd import kernels as K  # noqa: E402

SHAPES = [("down fwd", "nt", 4592, 3584, 18944), ("gate_up fwd", "nt", 4592, 37888, 3584), ("gate_up dX", "nn", 4592, 3584, 37888),
          ("down dW", "tn", 3584, 18944, 4592)]
for name, lay, m, n, k in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(1)
    r = lambda *s: (torch.rand(*s, device="cuda", generator=g) * 2 - 1).bfloat16()
    a, b, fn = (r(m, k), r(n, k), K.mm_nt) if lay == "nt" else ((r(m, k), r(k, n), K.mm_nn) if lay == "nn" else (r(k, m), r(k, n), K.mm_tn))
    out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        fn(a, b, out=out)
    torch.cuda.synchronize()
    w = out.view(-1)[:256].view(torch.int32).cpu().view(8, 16)
    nk = int(w[0, 8])
    print(f"{name} {lay} {m}x{n}x{k}: {nk} K tiles")
    for wave in (0, 4):
        iv = [int(x) / max(nk - (i == 7), 1) for i, x in enumerate(w[wave, :8])]
        print(f"  wave {wave} (group {wave >> 2}): " + " ".join(f"{x:6.0f}" for x in iv) + f"   sum {sum(iv):6.0f} cycles per K tile (MFMA floor 2048)")
