#!/usr/bin/env python
"""DB-pi0 fine-tune step and action inference at full size (BASELINE.json configs[3]: PaliGemma-3B-class backbone =
SigLIP-So400m/14@224 + Gemma-2B, 300 M action expert, 3 cameras, 48-token prompt, 16-step action chunk, action_dim 32),
synthetic data, random-init weights, bf16 compute / fp32 master.  One GPU:  python scripts/pi0_bench.py [steps] [batch]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    from dexbotic_amd.engine import OptimConfig
    from dexbotic_amd.model.pi0.pi0_arch import Pi0Config, Pi0ForCausalLM
    from dexbotic_amd.trainer import NativeTrainer
    dev = torch.device("cuda", 0)
    gem = dict(model_type="gemma")                                              # GemmaConfig defaults = Gemma-2B
    act = dict(model_type="gemma", hidden_size=1024, intermediate_size=4096)
    vis = dict(model_type="siglip_vision_model")                                # SiglipVisionConfig defaults = So400m/14
    chunk = 16
    cfg = Pi0Config(vision_config=vis, action_config=act, llm_config=gem, mm_projector_type="linear", action_dim=32,
                    chunk_size=chunk, compute_dtype="bfloat16")
    m = Pi0ForCausalLM(cfg, device=dev, train=True)
    m.init_random_(seed=0)
    for n in m.store.slots:                                                     # GemmaRMSNorm scales by (1 + w)
        if n.startswith(("model.llm.", "model.action_expert.")) and "norm" in n:
            m.store.w32(n).zero_()
    m.post_load()
    m.train()
    tr = NativeTrainer(m, OptimConfig(base_lr=2.5e-5, weight_decay=1e-10, adam_beta2=0.95, max_grad_norm=1.0), total_steps=1000)
    g = torch.Generator().manual_seed(1)
    batch = dict(input_ids=torch.randint(1000, 30000, (B, 48), generator=g).to(dev),
                 attention_mask=torch.ones(B, 48, dtype=torch.bool),
                 images=torch.randn(B, 3, 3, 224, 224, generator=g).clamp_(-2.5, 2.5).to(dev),
                 image_masks=torch.ones(B, 3, dtype=torch.bool), states=torch.randn(B, 32, generator=g).to(dev),
                 actions=torch.randn(B, chunk, 32, generator=g).to(dev))
    infer_only = os.environ.get("INFER_ONLY")                                   # per-request kernel tables: REQS requests, nothing else
    res = {"params_billion": round(m.store.total / 1e9, 3)}
    if not infer_only:
        for _ in range(2):
            loss = tr.step(batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = tr.step(batch)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        res = {"metric": "samples/sec DB-pi0 fine-tune", "value": round(B / dt, 2), "ms_per_step": round(1e3 * dt, 1),
               "batch": B, "loss": round(float(loss), 4), "params_billion": round(m.store.total / 1e9, 3)}
    if os.environ.get("SKIP_INFER"):
        print(json.dumps(res), flush=True)
        return
    m.eval()
    b1 = {k: v[:1] for k, v in batch.items() if k != "actions"}
    lat = []
    for _ in range(int(os.environ.get("REQS", "8"))):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        a = m.inference_action(diffusion_steps=10, **b1)
        a.cpu()
        lat.append(1e3 * (time.perf_counter() - t0))
    res["p50_action_inference_ms"] = round(float(np.median(lat[2:])), 1)
    res["inference_config"] = "bf16 compute, chunk 16, 10 Euler steps"
    if infer_only:
        print(json.dumps(res), flush=True)
        return
    # the way the REFERENCE serves pi0: fp32 weights and arithmetic (pi0_exp.py:347-353; matmul precision "highest", :106) and its
    # default chunk_size 50 (pi0_arch.py:58-59)
    del tr, m
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    cfg32 = Pi0Config(vision_config=vis, action_config=act, llm_config=gem, mm_projector_type="linear", action_dim=32,
                      chunk_size=50, compute_dtype="float32")
    m32 = Pi0ForCausalLM(cfg32, device=dev, train=False)
    m32.init_random_(seed=0)
    for n in m32.store.slots:
        if n.startswith(("model.llm.", "model.action_expert.")) and "norm" in n:
            m32.store.w32(n).zero_()
    m32.post_load()
    m32.eval()
    lat = []
    for _ in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        a = m32.inference_action(diffusion_steps=10, **b1)
        a.cpu()
        lat.append(1e3 * (time.perf_counter() - t0))
    res["p50_action_inference_ms_fp32_chunk50"] = round(float(np.median(lat[2:])), 1)
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
