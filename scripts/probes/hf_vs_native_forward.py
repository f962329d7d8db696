#!/usr/bin/env python
"""first training forward of the native model under NativeTrainer and under HF Trainer.training_step: where do they part?"""
import os, sys
import torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from tests.helpers import build_product, load_golden
from tests.test_hf_trainer_gpu import _batches, _args, STEPS, LR
from transformers import Trainer
from dexbotic_amd.engine import OptimConfig
from dexbotic_amd.trainer import NativeTrainer
from dexbotic_amd.model.cogact.action_model.dit import DiT

dtype = sys.argv[1] if len(sys.argv) > 1 else "bfloat16"
g, cfg, w = load_golden(os.path.join(R, "tests", "golden"), "t1")
batches = _batches(g, STEPS)
cap = {}
orig = DiT.forward


def hooked(self, x, t, z, *a, **k):
    out = orig(self, x, t, z, *a, **k)
    cap.setdefault(cur[0], []).append((x.detach().clone(), t.detach().clone(), z.detach().clone(), out.detach().clone()))
    return out


DiT.forward = hooked
cur = ["native"]
m = build_product(cfg, w, dtype, "cuda", train=True)
m.train()
tr = NativeTrainer(m, OptimConfig(base_lr=LR, weight_decay=0.0, max_grad_norm=1.0))
l0 = tr.step(batches[0]).item()
cur[0] = "hf"
m2 = build_product(cfg, w, dtype, "cuda", train=True)
t2 = Trainer(model=m2, args=_args(bf16=(dtype == "bfloat16")), train_dataset=[0] * 8)
t2.create_optimizer_and_scheduler(num_training_steps=STEPS)
t2.current_gradient_accumulation_steps = 1
m2.zero_grad()
l1 = float(t2.training_step(m2, batches[0]))
print("loss native", l0, "hf", l1)
for name, i in (("x_t", 0), ("t", 1), ("z", 2), ("eps_hat", 3)):
    a, b = cap["native"][0][i], cap["hf"][0][i]
    print(name, a.dtype, b.dtype, tuple(a.shape), "equal" if torch.equal(a, b) else f"DIFFER max {(a.float() - b.float()).abs().max().item():.3e}")
