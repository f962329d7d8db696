#!/usr/bin/env python
"""where one workgroup of the persistent decode step spends a token (tuning build: SRC=decode_fused.hip scripts/build_variant.sh decs<N>
-DDXA_DEC_STAMPS=<N>, DXA_LIB=_abl/lib_decs<N>.so): cycles of work and of barrier wait per phase, per token"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from dexbotic_amd import _lib as L  # noqa: E402
from dexbotic_amd.model.dexbotic_arch import DexboticConfig, DexboticForCausalLM  # noqa: E402
from dexbotic_amd.model.llm.qwen2 import Qwen2Config  # noqa: E402
from dexbotic_amd.model.modules.mm_vision.clip.clip_encoder import CLIPVisionConfig  # noqa: E402
dev = torch.device("cuda", 0)
m = DexboticForCausalLM(DexboticConfig(llm_config=Qwen2Config(), mm_vision_tower=CLIPVisionConfig(), mm_projector_type="mlp2x_gelu",
                                       compute_dtype="bfloat16"), device=dev, train=False)
m.init_random_(seed=0)
m.eval()
b = bench.synthetic_batch(1, 1, 32, dev, seed=3)
lib = ctypes.CDLL(L.LIB_PATH)
n = 33
m.generate(b["input_ids"], images=b["images"], max_new_tokens=n)
st = (ctypes.c_ulonglong * 16)()
lib.dxa_decode_debug_stamps(st)
wg = (ctypes.c_ulonglong * (5 * 256))()
lib.dxa_decode_debug_wg(wg)
m.generate(b["input_ids"], images=b["images"], max_new_tokens=n)
lib.dxa_decode_debug_stamps(st)
lib.dxa_decode_debug_wg(wg)
tok = n - 1
names = ["qkv", "attention", "o", "gate/up", "down"]
tot = sum(st[:10])
print(f"per token, cycles of the 100 MHz-or-shader counter (total {tot / tok:.0f}):")
for i, nm in enumerate(names):
    print(f"  {nm:10s} work {st[2 * i] / tok / 28:9.0f}  barrier {st[2 * i + 1] / tok / 28:9.0f}   per layer; share of the token {100.0 * (st[2 * i] + st[2 * i + 1]) / tot:5.1f} %")
print(f"  attention, per layer: request + RoPE {st[10] / tok / 28:.0f}  keys {st[11] / tok / 28:.0f}  merge + store {st[12] / tok / 28:.0f}")
import numpy as np
w = np.array(list(wg), dtype=np.float64).reshape(5, 256) / tok / 28
print("work cycles per layer over the 256 workgroups (min / median / max; the slowest 8 with their index; mean per index % 8 = XCD):")
for i, nm in enumerate(names):
    r = w[i]
    slow = np.argsort(r)[-8:][::-1]
    print(f"  {nm:10s} {r.min():8.0f} {np.median(r):8.0f} {r.max():8.0f}   slowest {[(int(j), int(r[j])) for j in slow]}")
    print(f"             per XCD {[int(r[x::8].mean()) for x in range(8)]}")
