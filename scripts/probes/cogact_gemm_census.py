#!/usr/bin/env python
"""which fp32 products of a DB-CogACT training step (2 decoder layers, the full DiT-B head) do NOT take the split-bf16 (bf16x3) path:
shape census of kernels.gemm calls with fp32 operands"""
import collections, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from dexbotic_amd import kernels as K  # noqa: E402
from dexbotic_amd.engine import OptimConfig  # noqa: E402
from dexbotic_amd.model.llm.qwen2 import Qwen2Config  # noqa: E402
from dexbotic_amd.model.cogact.cogact_arch import CogActConfig, CogACTForCausalLM  # noqa: E402
from dexbotic_amd.model.modules.mm_vision.clip.clip_encoder import CLIPVisionConfig  # noqa: E402
from dexbotic_amd.trainer import NativeTrainer  # noqa: E402
dev = torch.device("cuda", 0)
cfg = CogActConfig(llm_config=Qwen2Config(num_hidden_layers=2), mm_vision_tower=CLIPVisionConfig(num_hidden_layers=3), mm_projector_type="mlp2x_gelu",
                   action_model_type="DiT-B", action_dim=7, chunk_size=16, compute_dtype="bfloat16")
m = CogACTForCausalLM(cfg, device=dev, train=True)
m.init_random_(seed=0)
m.train()
tr = NativeTrainer(m, OptimConfig(base_lr=2e-5, weight_decay=0.0, max_grad_norm=1.0), total_steps=1000)
batch = bench.synthetic_batch(16, 1, 32, dev, seed=5)
batch.pop("labels", None)
tr.step(batch)
census = collections.Counter()
orig = K.gemm
def spy(layout, a, b, M, N, Kc, lda, ldb, out, ldc, **kw):
    if a.dtype == torch.float32:
        x3 = kw.get("epi_f32") or K._x3_eligible(layout, a, b, out, M, N, Kc, lda, ldb, kw.get("nb", (1, 1, 1)))
        census[("x3" if x3 else "exact", {0: "NT", 1: "NN", 2: "TN"}[layout], M, N, Kc, tuple(kw.get("nb", (1, 1, 1))))] += 1
    return orig(layout, a, b, M, N, Kc, lda, ldb, out, ldc, **kw)
K.gemm = spy
tr.step(batch)
torch.cuda.synchronize()
for k, v in sorted(census.items(), key=lambda kv: (kv[0][0], -kv[1] * kv[0][2] * kv[0][3] * kv[0][4])):
    print(v, k, f"{2e-9 * k[2] * k[3] * k[4] * v * (k[5][0] * k[5][1] * k[5][2]):.2f} GF")
