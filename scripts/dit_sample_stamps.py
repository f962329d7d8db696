#!/usr/bin/env python
"""Segment times of the one-launch sampler's phases (tuning build: SRC=dit_fused.hip scripts/build_variant.sh ditstamp0
-DDXA_DIT_STAMPS=0; run with DXA_LIB=_abl/lib_ditstamp0.so; DXA_DIT_BF16=0 for the exact-fp32 kernel).  Average over the 120 block
passes of one 10-step sample."""
import ctypes
import os
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dexbotic_amd import _lib as L  # noqa: E402
from dexbotic_amd.engine import ParamStore, attach_parameters, building  # noqa: E402
from dexbotic_amd.model.cogact.action_model.builder import build_action_model  # noqa: E402

dev = "cuda"
st = ParamStore(dev, torch.bfloat16)
with building(st):
    head = build_action_model(types.SimpleNamespace(action_model_type="DiT-B", hidden_size=3584, action_dim=7, chunk_size=16))
st.finalize(train=False)
attach_parameters(torch.nn.Module(), st)
st.master.normal_(0.0, 0.02, generator=torch.Generator(device=dev).manual_seed(0))
head.eval()
head.create_ddim(10)
z, noise = torch.randn(2, 1, 3584, device=dev), torch.randn(1, 16, 7, device=dev)
with torch.no_grad():
    for _ in range(4):
        head.net.ddim_sample_fused(noise, z, head.ddim_diffusion, 1.5)
torch.cuda.synchronize()
lib = ctypes.CDLL(L.LIB_PATH)
out = (ctypes.c_ulonglong * 40)()
assert lib.dxa_dit_debug_stamps(out) == 0
names = ["qkv", "attention", "proj", "fc1", "fc2"]
print("cycles per phase; segments: operands+MFMA | partials in LDS | fold+epilogue | ack+assemble | barrier")
tot = 0.0
for i, n in enumerate(names):
    c = max(out[i * 8 + 7], 1)
    v = [out[i * 8 + j] / c for j in range(5)]
    tot += sum(v)
    print(f"  {n:10s} " + " ".join(f"{x:8.0f}" for x in v) + f"   sum {sum(v):8.0f}   ({c} phases)")
print(f"  per block {tot:8.0f} cycles")
