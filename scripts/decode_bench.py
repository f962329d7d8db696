#!/usr/bin/env python
"""Autoregressive decode of the full-size discrete VLA (Qwen2.5-7B-class decoder + CLIP-L/14 + lm_head, bf16):
prefill latency and ms per generated token over a KV cache (BASELINE.json configs[0] shape: single image, greedy)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    from dexbotic_amd.model.dexbotic_arch import DexboticConfig, DexboticForCausalLM
    from dexbotic_amd.model.llm.qwen2 import Qwen2Config
    from dexbotic_amd.model.modules.mm_vision.clip.clip_encoder import CLIPVisionConfig
    dev = torch.device("cuda", 0)
    cfg = DexboticConfig(llm_config=Qwen2Config(), mm_vision_tower=CLIPVisionConfig(), mm_projector_type="mlp2x_gelu",
                         compute_dtype="bfloat16")
    m = DexboticForCausalLM(cfg, device=dev, train=False)
    m.init_random_(seed=0)
    m.eval()
    b = bench.synthetic_batch(1, 1, 32, dev, seed=3)
    n_new = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m.generate(b["input_ids"], images=b["images"], max_new_tokens=1)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        m.generate(b["input_ids"], images=b["images"], max_new_tokens=n_new)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        pre = 1e3 * (t1 - t0)
        per = 1e3 * ((t2 - t1) - (t1 - t0)) / (n_new - 1)
        print(f"rep {rep}: prefill+1 token {pre:.1f} ms, decode {per:.2f} ms/token "
              f"({15.2e9 * 1e-9 / per:.2f} TB/s of weight streaming)", flush=True)


if __name__ == "__main__":
    main()
