#!/usr/bin/env python
"""MemVLA fine-tune step and per-frame action inference at full size (BASELINE.json configs[4]: Qwen2.5-7B-class decoder
+ CLIP-L/14@224 + DiT-L with perceptual attention, per_token_size 256, memory of 4 past frames), 'group' batches of
16 consecutive frames of one episode, synthetic data, bf16 compute / fp32 master.   python scripts/memvla_bench.py [steps]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    B = 16
    from dexbotic_amd.engine import OptimConfig
    from dexbotic_amd.model.llm.qwen2 import Qwen2Config
    from dexbotic_amd.model.memvla.memvla_arch import MemVLAConfig, MemVLAForCausalLM
    from dexbotic_amd.model.modules.mm_vision.clip.clip_encoder import CLIPVisionConfig
    from dexbotic_amd.trainer import NativeTrainer
    dev = torch.device("cuda", 0)
    cfg = MemVLAConfig(llm_config=Qwen2Config(), mm_vision_tower=CLIPVisionConfig(), mm_projector_type="mlp2x_gelu",
                       action_model_type="DiT-L", action_dim=7, chunk_size=16, compute_dtype="bfloat16", per_token_size=256,
                       dataloader_type="group", group_size=16, mem_length=4, retrieval_layers=2, use_timestep_pe=True,
                       fusion_type="gate", consolidate_type="tome")
    m = MemVLAForCausalLM(cfg, device=dev, train=True)
    m.init_random_(seed=0)
    m.train()
    if os.environ.get("SKIP_TRAIN"):
        return infer(m, dev, {"params_billion": round(m.store.total / 1e9, 3)})
    tr = NativeTrainer(m, OptimConfig(base_lr=2e-5, weight_decay=0.0, max_grad_norm=1.0), total_steps=1000)
    batch = bench.synthetic_batch(B, 1, 32, dev, seed=5)
    batch.pop("labels")
    batch["indexes"] = [[0, 3, 100 + i] for i in range(B)]
    for _ in range(2):
        loss = tr.step(batch)
    torch.cuda.synchronize()
    from dexbotic_amd import hostcpu
    th0 = hostcpu.throttle_stats()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = tr.step(batch)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    th1 = hostcpu.throttle_stats()
    res = {"metric": "samples/sec MemVLA fine-tune", "value": round(B / dt, 2), "ms_per_step": round(1e3 * dt, 1), "batch": B,
           "loss": round(float(loss), 4), "params_billion": round(m.store.total / 1e9, 3), "host_threads": tr.host_threads,
           "cgroup_throttled_periods": th1.get("nr_throttled", 0) - th0.get("nr_throttled", 0),
           "cgroup_periods": th1.get("nr_periods", 0) - th0.get("nr_periods", 0)}
    if os.environ.get("SKIP_INFER"):
        print(json.dumps(res), flush=True)
        return
    infer(m, dev, res)


def infer(m, dev, res):
    m.eval()
    b1 = bench.synthetic_batch(1, 1, 32, dev, seed=7)
    norms = {"min": [-1.0] * 7, "max": [1.0] * 7}
    lat = []
    extra = {"cache_per_kv": False} if os.environ.get("NO_KV_CACHE") else {}
    for f in range(int(os.environ.get("FRAMES", "12"))):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m.inference_action(b1["input_ids"], b1["images"], "True" if f == 0 else "False",
                           {"cfg_scale": 1.5, "num_ddim_steps": 10, "action_norms": norms, **extra})
        lat.append(1e3 * (time.perf_counter() - t0))
    res["p50_frame_inference_ms"] = round(float(np.median(lat[4:])), 1)
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
