#!/usr/bin/env python
"""Segment times of the persistent DiT-block kernel's phases (tuning build: SRC=dit_fused.hip scripts/build_variant.sh ditstamp
-DDXA_DIT_STAMPS=<workgroup>; run with DXA_LIB=_abl/lib_ditstamp.so).  Per phase type, the average over the 12 blocks of one launch."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dexbotic_amd import kernels as K  # noqa: E402
from dexbotic_amd import _lib as L  # noqa: E402

N, T1, H, heads, I, depth = 2, 17, 768, 12, 3072, 12
g = torch.Generator().manual_seed(0)
r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to("cuda")
ptrs, keep = [], []
for _ in range(depth):
    blk = [r(3 * H, H, sc=H ** -0.5), r(3 * H, sc=0.1), r(H, H, sc=H ** -0.5), r(H, sc=0.1), r(I, H, sc=H ** -0.5), r(I, sc=0.1),
           r(H, I, sc=I ** -0.5), r(H, sc=0.1)]
    keep.append(blk)
    ptrs += [w.data_ptr() for w in (keep[0] if os.environ.get("DIT_SAME_W") else blk)]     # DIT_SAME_W=1: every block reads block 0's weights (28 MB: warm in the caches / TLBs)
table = torch.tensor(ptrs, dtype=torch.int64).to("cuda")
h0 = r(N * T1, H)
for _ in range(5):
    K.dit_blocks_fwd(h0.clone(), table, depth, N, T1, H, heads, I, 1e-6)
torch.cuda.synchronize()
lib = ctypes.CDLL(L.LIB_PATH)
out = (ctypes.c_ulonglong * 40)()
assert lib.dxa_dit_debug_stamps(out) == 0
names = ["qkv", "attention", "proj", "fc1", "fc2"]
seg = ["operands+MFMA", "partials in LDS", "fold+epilogue", "ack+assemble", "barrier"]
print("cycles per phase (one launch, average over the blocks); segments: " + " | ".join(seg))
tot = 0.0
for i, n in enumerate(names):
    c = max(out[i * 8 + 7], 1)
    v = [out[i * 8 + j] / c for j in range(5)]
    tot += sum(v)
    print(f"  {n:10s} " + " ".join(f"{x:8.0f}" for x in v) + f"   sum {sum(v):8.0f}")
print(f"  per block {tot:8.0f} cycles")
