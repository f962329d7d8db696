"""MemVLA golden vectors at the REAL size from the reference's own MemVLAForCausalLM — TEST INFRASTRUCTURE.

    python -m oracle.gen_golden_memvla_real      # build container only (needs /root/reference) -> tests/golden/memvla_real_ref.npz

BASELINE.json configs[4] shapes: Qwen2.5-7B-class decoder widths (d 3584, 28 q / 4 kv x 128, ffn 18944; ONE layer), CLIP-L/14
@224 (2 used + 1 unused layers), per_token_size 256, **DiT-L (24 blocks, 1024-d, 16 heads) with the perceptual cross
attention in every block**, memory of 4 past frames ('tome' consolidation above that), 2 retrieval layers with timestep PE and
gate fusion.  One 'group' batch of 6 consecutive frames of one episode: the bank grows to its depth of 4 and consolidates
twice.  Retrieval dropout off (the deterministic configuration; memvla_drop_t1.npz pins the dropout path at the tiny size).
fp32 training step AND the same step under bf16 autocast (loss, per-group gradient norms, strided gradient samples; their
relative distances under ref_bf16_vs_fp32/*) + a 5-frame fp32 inference episode.
tests/golden/memvla_t1.npz (hidden 256, DiT 3 x 128) reaches neither the MFMA tile widths nor DiT-L's per_attn shapes."""
from __future__ import annotations

import os
import sys
import time
import zlib

import numpy as np
import torch

from . import gen_golden_memvla as GM
from .cogact_oracle import OracleConfig
from .memvla_oracle import memvla_shapes
from .weights import make_weights, weights_crc

REAL = OracleConfig(vocab_size=2048, hidden_size=3584, intermediate_size=18944, num_hidden_layers=1,
                    num_attention_heads=28, num_key_value_heads=4, v_hidden=1024, v_inter=4096, v_layers=3, v_heads=16,
                    v_image=224, v_patch=14, dit_hidden=1024, dit_depth=24, dit_heads=16)
PER, MEM_LEN, GROUP = 256, 4, 6
SEED = 977
STRIDE = 499
GROUPS = {"llm": "model.llm.", "vision": "model.mm_vision_tower.", "projector": "model.mm_projector.",
          "head": "model.action_head.", "bank": "model.per_cog_mem_bank.", "compr": "model.per_compr."}
GSAMP = ("model.per_compr.reduce.2.weight", "model.per_cog_mem_bank.retrieval_blocks.cog.1.q_proj.weight",
         "model.per_cog_mem_bank.retrieval_blocks.per.0.ffn.3.weight", "model.per_cog_mem_bank.gate_fusion_blocks.per.proj.weight",
         "model.action_head.net.blocks.23.per_attn.in_proj_weight", "model.action_head.net.blocks.0.mlp.fc1.weight",
         "model.action_head.net.blocks.11.norm3.weight", "model.llm.layers.0.mlp.down_proj.weight", "model.mm_projector.2.weight")


def inputs(cfg=REAL):
    rs = np.random.RandomState(19)
    B, L = GROUP, 32
    ids = rs.randint(10, cfg.vocab_size - 10, size=(B, L)).astype(np.int64)
    ids[:, 1] = -200
    mask = np.ones((B, L), dtype=bool)
    mask[2, 27:] = False
    images = np.clip(rs.standard_normal((B, 3, cfg.v_image, cfg.v_image)), -2.5, 2.5).astype(np.float32)
    actions = rs.uniform(-1, 1, size=(B, cfg.chunk_size * cfg.action_dim)).astype(np.float32)
    indexes = np.array([[0, 5, 200 + i] for i in range(B)], dtype=np.int64)
    R = 4
    noise = rs.standard_normal((R * B, cfg.chunk_size, cfg.action_dim)).astype(np.float32)
    timesteps = rs.randint(0, cfg.diffusion_steps, size=(R * B,)).astype(np.int64)
    drop_u = rs.uniform(0, 1, size=(R * B,)).astype(np.float32)
    drop_u[5] = 0.01
    frames = np.clip(rs.standard_normal((5, 3, cfg.v_image, cfg.v_image)), -2.5, 2.5).astype(np.float32)
    inits = rs.standard_normal((5, 1, cfg.chunk_size, cfg.action_dim)).astype(np.float32)
    return dict(input_ids=ids, attention_mask=mask, images=images, actions=actions, indexes=indexes, noise=noise,
                timesteps=timesteps, drop_u=drop_u, infer_frames=frames, infer_inits=inits, infer_prompt=ids[:1].copy())


def summarize(grads, loss):
    res = {"loss": np.float64(loss)}
    for g, pre in GROUPS.items():
        sq = sum(float(v.double().pow(2).sum()) for n, v in grads.items() if v is not None and n.startswith(pre))
        res[f"gnorm/{g}"] = np.float64(sq ** 0.5)
    for n in GSAMP:
        res["gsamp/" + n] = grads[n].reshape(-1)[::STRIDE].float().cpu().numpy().copy()
    return res


def main():
    sys.path.insert(0, GM.REF)
    sys.path.insert(0, GM.ROOT)
    torch.set_num_threads(os.cpu_count() or 8)
    from .gen_golden import inject_rng, install_timm_shim
    install_timm_shim()
    t0 = time.time()
    w = make_weights(memvla_shapes(REAL, PER), SEED)
    m = GM.build_reference(REAL, w, per=PER, mem_len=MEM_LEN, group_size=GROUP)
    x = inputs()
    t = torch.from_numpy
    res = {"seed": np.int64(SEED), "weights_crc": np.int64(weights_crc(w)), "per_token_size": np.int64(PER),
           "mem_length": np.int64(MEM_LEN), "images_crc": np.int64(zlib.crc32(x["images"].tobytes())),
           "frames_crc": np.int64(zlib.crc32(x["infer_frames"].tobytes()))}
    res.update({k: v for k, v in x.items() if k not in ("images", "infer_frames")})
    m.train()
    # "fp32": the plain float32 step.  "bf16" (round 4): the same step under torch.autocast("cpu", bfloat16) with fp32 weights — what
    # HF Trainer does for bf16=True (exp/trainer.py:100-124) — with the ONE shim of gen_golden_realwidth_ref.py: the reference wraps
    # the action-head loss in autocast('cuda', float32) (memvla_arch.py:647), which on its real device switches autocast off for the
    # head; that context manager does not touch the CPU autocast state, so the head is taken out of it here (fp32 inputs).
    head = m.model.action_head_module
    head_loss = head.loss

    def fp32_head_loss(xx, zz, *a, **k):
        with torch.autocast("cpu", enabled=False):
            return head_loss(xx.float(), zz.float(), *a, **{n: (v.float() if torch.is_tensor(v) else v) for n, v in k.items()})
    for tag, ac in (("fp32", False), ("bf16", True)):
        m.zero_grad(set_to_none=True)
        head.loss = fp32_head_loss
        try:
            with inject_rng(noise=t(x["noise"]), timesteps=t(x["timesteps"]), drop_u=t(x["drop_u"])):
                with torch.autocast("cpu", dtype=torch.bfloat16, enabled=ac):
                    out = m(input_ids=t(x["input_ids"]), attention_mask=t(x["attention_mask"]), images=t(x["images"]),
                            actions=t(x["actions"]), indexes=[list(map(int, r)) for r in x["indexes"]])
        finally:
            head.loss = head_loss
        out.loss.backward()
        r = summarize({n: p.grad for n, p in m.named_parameters()}, out.loss.item())
        print(tag, "loss", float(r["loss"]), {k: round(float(v), 5) for k, v in r.items() if k.startswith("gnorm/")},
              f"{time.time()-t0:.0f}s", flush=True)
        for k, v in r.items():
            res[tag + "/" + k] = v
    for k in [k for k in res if k.startswith("bf16/")]:
        a, b = np.asarray(res[k], dtype=np.float64), np.asarray(res["fp32/" + k[5:]], dtype=np.float64)
        res["ref_bf16_vs_fp32/" + k[5:]] = np.float64(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
    m.zero_grad(set_to_none=True)
    m.eval()
    norms = {"min": [-1.0] * REAL.action_dim, "max": [1.0] * REAL.action_dim}
    acts = []
    for f in range(5):
        with torch.no_grad(), inject_rng(init_noise=t(x["infer_inits"][f])):
            a = m.inference_action(t(x["infer_prompt"]), t(x["infer_frames"][f:f + 1]), "True" if f == 0 else "False",
                                   {"cfg_scale": 1.5, "num_ddim_steps": 10, "action_norms": norms})
        acts.append(np.array(a, dtype=np.float32))
    res["fp32/infer_actions"] = np.stack(acts)
    print("infer |a|", float(np.abs(res["fp32/infer_actions"]).mean()), f"{time.time()-t0:.0f}s", flush=True)
    dst = os.path.join(GM.GOLD, "memvla_real_ref.npz")
    np.savez_compressed(dst, **res)
    print("wrote", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
