#!/usr/bin/env python
"""Which torch (aten) ops — i.e. launches that are NOT libdexbotic_amd kernels — run inside one MemVLA (default) or CogACT
training step, by call site inside dexbotic_amd/.  The per-step kernel tables (profiles/*_per_step_kernel_stats.txt) show the
at::native kernels; this tells where they come from.   python scripts/torch_op_census.py [memvla|cogact]"""
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

VIEW = ("view", "reshape", "permute", "transpose", "expand", "slice", "select", "unsqueeze", "squeeze", "as_strided", "alias",
        "detach", "t.default", "unbind", "split", "_unsafe_view", "size", "stride", "is_", "sym_", "dim", "numel", "_local_scalar",
        "lift_fresh", "empty", "set_", "narrow", "unfold", "chunk", "contiguous")


class Census(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.counts = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(v in name for v in VIEW):
            site = "?"
            for fr in reversed(traceback.extract_stack(limit=18)[:-1]):
                if "dexbotic_amd" in fr.filename and "kernels.py" not in fr.filename:
                    site = f"{os.path.relpath(fr.filename)}:{fr.lineno}"
                    break
            self.counts[(name.replace("aten.", ""), site)] += 1
        return func(*args, **(kwargs or {}))


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "memvla"
    dev = torch.device("cuda", 0)
    from dexbotic_amd.engine import OptimConfig
    from dexbotic_amd.trainer import NativeTrainer
    if what == "memvla":
        from dexbotic_amd.model.llm.qwen2 import Qwen2Config
        from dexbotic_amd.model.memvla.memvla_arch import MemVLAConfig, MemVLAForCausalLM
        from dexbotic_amd.model.modules.mm_vision.clip.clip_encoder import CLIPVisionConfig
        cfg = MemVLAConfig(llm_config=Qwen2Config(num_hidden_layers=2), mm_vision_tower=CLIPVisionConfig(num_hidden_layers=3),
                           mm_projector_type="mlp2x_gelu", action_model_type="DiT-L", action_dim=7, chunk_size=16,
                           compute_dtype="bfloat16", per_token_size=256, dataloader_type="group", group_size=16, mem_length=4,
                           retrieval_layers=2, use_timestep_pe=True, fusion_type="gate", consolidate_type="tome")
        m = MemVLAForCausalLM(cfg, device=dev, train=True)
        batch = bench.synthetic_batch(16, 1, 32, dev, seed=5)
        batch.pop("labels")
        batch["indexes"] = [[0, 3, 100 + i] for i in range(16)]
    else:
        class A:
            llm_layers, vit_layers, dtype = 2, 3, "bfloat16"
        m, *_ = bench.build_model(A, dev)
        batch = bench.synthetic_batch(16, 1, 32, dev, seed=5)
    m.init_random_(seed=0)
    m.train()
    tr = NativeTrainer(m, OptimConfig(base_lr=2e-5, weight_decay=0.0, max_grad_norm=1.0), total_steps=1000)
    tr.step(batch)
    torch.cuda.synchronize()
    c = Census()
    with c:
        tr.step(batch)
    torch.cuda.synchronize()
    tot = sum(c.counts.values())
    print(f"# {what}: {tot} non-view aten ops in one training step (2 decoder + 3 ViT layers; the per-layer ops scale with depth)")
    for (op, site), n in c.counts.most_common(90):
        print(f"{n:6d}  {op:42s} {site}")


if __name__ == "__main__":
    main()
