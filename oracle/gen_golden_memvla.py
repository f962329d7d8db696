"""Generate tests/golden/memvla_t1.npz by running the REFERENCE's MemVLAForCausalLM on CPU — TEST INFRASTRUCTURE.

    python -m oracle.gen_golden_memvla             # build container only (needs /root/reference)
    python -m oracle.gen_golden_memvla --dropout   # tests/golden/memvla_drop_t1.npz: the same training batch with the retrieval
                                                   # blocks' dropout 0.1 LEFT ON and every mask injected (MaskFeed)

A 'group'-mode training batch of 2 episodes x 3 consecutive frames with mem_length 2 (so the token-merge consolidation
runs), injected diffusion draws, loss + gradients; then a 4-frame inference episode through inference_action.  The
retrieval blocks' dropout (0.1, passed to SDPA even in eval) is set to 0 — see oracle/memvla_oracle.py.
"""
from __future__ import annotations

import os
import sys
import tempfile

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
PER, MEM_LEN = 32, 2


class MaskFeed:
    """the k-th dropout mask requested, as 0 | 1/(1-p): seeded by (seed, k) and the element count only, so the reference (masks
    of shape [B,h,N,M] / [B,N,4D] / [B,N,D]) and the product (same element order, other shapes) draw identical masks as long
    as they ask in the same order"""

    def __init__(self, seed: int, p: float):
        self.seed, self.p, self.k = int(seed), float(p), 0

    def __call__(self, shape):
        n = int(np.prod(shape))
        keep = np.random.default_rng([self.seed, self.k]).random(n) >= self.p
        self.k += 1
        return (keep.astype(np.float32) / np.float32(1.0 - self.p)).reshape(shape)


class inject_dropout:
    """while active, torch's SDPA with dropout_p > 0 and F.dropout with p > 0 (training) take their masks from `feed`"""

    def __init__(self, feed):
        self.feed = feed

    def __enter__(self):
        import torch.nn.functional as F
        self.F, self.sdpa, self.drop = F, F.scaled_dot_product_attention, F.dropout
        feed, sdpa0, drop0 = self.feed, self.sdpa, self.drop

        def sdpa(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False, scale=None, **kw):
            if not dropout_p:
                return sdpa0(q, k, v, attn_mask=attn_mask, dropout_p=0.0, is_causal=is_causal, scale=scale, **kw)
            assert attn_mask is None and not is_causal
            w = torch.softmax((q @ k.transpose(-1, -2)) * (scale if scale is not None else q.shape[-1] ** -0.5), dim=-1)
            return (w * torch.from_numpy(feed(tuple(w.shape))).to(w.dtype)) @ v

        def dropout(x, p=0.5, training=True, inplace=False):
            if not p or not training:
                return x
            return x * torch.from_numpy(feed(tuple(x.shape))).to(x.dtype)
        F.scaled_dot_product_attention, F.dropout = sdpa, dropout
        return self

    def __exit__(self, *exc):
        self.F.scaled_dot_product_attention, self.F.dropout = self.sdpa, self.drop
        return False


def build_reference(cfg, weights, keep_dropout=False, per=None, mem_len=None, group_size=3):
    from transformers import CLIPImageProcessor, CLIPVisionConfig, CLIPVisionModel, Qwen2Config
    from dexbotic.model.memvla.action_model import action_models
    from dexbotic.model.memvla.action_model.dit import DiT
    from dexbotic.model.memvla.memvla_arch import MemVLAConfig, MemVLAForCausalLM
    d = os.path.join(tempfile.mkdtemp(), "tiny_clip")
    vcfg = CLIPVisionConfig(hidden_size=cfg.v_hidden, intermediate_size=cfg.v_inter, num_hidden_layers=cfg.v_layers,
                            num_attention_heads=cfg.v_heads, image_size=cfg.v_image, patch_size=cfg.v_patch,
                            layer_norm_eps=cfg.v_eps)
    CLIPVisionModel(vcfg).save_pretrained(d)
    CLIPImageProcessor(size={"shortest_edge": cfg.v_image},
                       crop_size={"height": cfg.v_image, "width": cfg.v_image}).save_pretrained(d)
    action_models.DiT_models["DiT-T"] = lambda **kw: DiT(depth=cfg.dit_depth, hidden_size=cfg.dit_hidden,
                                                         num_heads=cfg.dit_heads, **kw)
    llm = Qwen2Config(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                      num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                      num_key_value_heads=cfg.num_key_value_heads, max_position_embeddings=4096,
                      rope_theta=cfg.rope_theta, rms_norm_eps=cfg.rms_norm_eps)
    c = MemVLAConfig(llm_config=llm, mm_vision_tower=d, mm_projector_type="mlp2x_gelu", action_model_type="DiT-T",
                     action_dim=cfg.action_dim, chunk_size=cfg.chunk_size, per_token_size=per or PER, dataloader_type="group",
                     group_size=group_size, mem_length=mem_len or MEM_LEN, retrieval_layers=2, use_timestep_pe=True, fusion_type="gate",
                     consolidate_type="tome")
    m = MemVLAForCausalLM(c)
    from oracle.memvla_oracle import memvla_shapes
    ref_shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    mine = memvla_shapes(cfg, per or PER)
    assert ref_shapes == mine, (set(ref_shapes) ^ set(mine))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()}, strict=True)
    for mod in m.modules():                                  # deterministic retrieval: no dropout anywhere
        if keep_dropout:
            break
        if hasattr(mod, "dropout") and isinstance(getattr(mod, "dropout"), float):
            mod.dropout = 0.0
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    for p_ in m.model.parameters():
        p_.requires_grad = True
    return m


def main():
    sys.path.insert(0, REF)
    sys.path.insert(0, ROOT)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    from oracle.gen_golden import inject_rng, install_timm_shim
    install_timm_shim()
    from oracle.cogact_oracle import OracleConfig
    from oracle.memvla_oracle import memvla_shapes
    from oracle.weights import make_weights, weights_crc
    cfg = OracleConfig()
    seed = 1357
    w = make_weights(memvla_shapes(cfg, PER), seed)
    m = build_reference(cfg, w)
    m.train()
    rs = np.random.RandomState(seed + 1)
    B, L = 6, 12
    ids = rs.randint(10, cfg.vocab_size - 10, size=(B, L)).astype(np.int64)
    ids[:, 1] = -200
    mask = np.ones((B, L), dtype=bool)
    mask[1, 10:] = False
    mask[4, 9:] = False
    images = np.clip(rs.standard_normal((B, 3, cfg.v_image, cfg.v_image)), -2.5, 2.5).astype(np.float32)
    actions = rs.uniform(-1, 1, size=(B, cfg.chunk_size * cfg.action_dim)).astype(np.float32)
    indexes = np.array([[0, 7, 10], [0, 7, 11], [0, 7, 12], [1, 3, 40], [1, 3, 41], [1, 3, 42]], dtype=np.int64)
    R = 4
    noise = rs.standard_normal((R * B, cfg.chunk_size, cfg.action_dim)).astype(np.float32)
    timesteps = rs.randint(0, cfg.diffusion_steps, size=(R * B,)).astype(np.int64)
    drop_u = rs.uniform(0, 1, size=(R * B,)).astype(np.float32)
    drop_u[3] = 0.01
    t = torch.from_numpy
    with inject_rng(noise=t(noise), timesteps=t(timesteps), drop_u=t(drop_u)):
        out = m(input_ids=t(ids), attention_mask=t(mask), images=t(images), actions=t(actions),
                indexes=[list(map(int, r)) for r in indexes])
    out.loss.backward()
    sd = dict(m.named_parameters())
    res = dict(weights_crc=np.uint32(weights_crc(w)), seed=np.int64(seed), input_ids=ids, attention_mask=mask, images=images,
               actions=actions, indexes=indexes, noise=noise, timesteps=timesteps, drop_u=drop_u,
               loss=np.float32(out.loss.item()), per_token_size=np.int64(PER), mem_length=np.int64(MEM_LEN))
    gsq = 0.0
    nograd = []
    for n, p_ in sd.items():
        if p_.grad is not None:
            gsq += float(p_.grad.double().pow(2).sum())
            res["gradN/" + n] = np.float64(p_.grad.double().norm().item())
        else:
            nograd.append(n)
    res["grad_norm"] = np.float64(gsq ** 0.5)
    for n in ("model.per_compr.reduce.2.weight", "model.per_cog_mem_bank.retrieval_blocks.cog.1.q_proj.weight",
              "model.per_cog_mem_bank.gate_fusion_blocks.per.proj.weight",
              "model.per_cog_mem_bank.timestep_embedders.per.mlp.0.weight",
              "model.action_head.net.blocks.1.per_attn.in_proj_weight", "model.action_head.net.blocks.0.norm3.bias",
              "model.llm.layers.0.self_attn.q_proj.weight", "model.mm_projector.2.weight"):
        res["grad/" + n] = sd[n].grad.numpy().astype(np.float32)
    print("[gen_golden_memvla] no grad:", nograd)
    # inference: one 4-frame episode (memory reset on the first frame), injected initial noise per frame
    m.eval()
    norms = {"min": [-1.0] * cfg.action_dim, "max": [1.0] * cfg.action_dim}
    frames = np.clip(rs.standard_normal((4, 3, cfg.v_image, cfg.v_image)), -2.5, 2.5).astype(np.float32)
    inits = rs.standard_normal((4, 1, cfg.chunk_size, cfg.action_dim)).astype(np.float32)
    prompt = ids[:1].copy()
    acts = []
    for f in range(4):
        with torch.no_grad(), inject_rng(init_noise=t(inits[f])):
            a = m.inference_action(t(prompt), t(frames[f:f + 1]), "True" if f == 0 else "False",
                                   {"cfg_scale": 1.5, "num_ddim_steps": 10, "action_norms": norms})
        acts.append(np.array(a, dtype=np.float32))
    res.update(infer_prompt=prompt, infer_frames=frames, infer_inits=inits, infer_actions=np.stack(acts))
    np.savez_compressed(os.path.join(GOLD, "memvla_t1.npz"), **res)
    print(f"[gen_golden_memvla] loss {res['loss']:.6f} |g| {res['grad_norm']:.4f} infer |a| {np.abs(res['infer_actions']).mean():.4f}")


def main_dropout():
    """the training batch of memvla_t1.npz through the reference with its retrieval dropout (0.1: SDPA weights + the two
    nn.Dropout of the FFN) ON, masks injected by MaskFeed(seed 2468): loss, every gradient norm, a few gradients"""
    sys.path.insert(0, REF)
    sys.path.insert(0, ROOT)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    from oracle.gen_golden import inject_rng, install_timm_shim
    install_timm_shim()
    from oracle.cogact_oracle import OracleConfig
    from oracle.memvla_oracle import memvla_shapes
    from oracle.weights import make_weights, weights_crc
    cfg = OracleConfig()
    g = np.load(os.path.join(GOLD, "memvla_t1.npz"))
    w = make_weights(memvla_shapes(cfg, PER), int(g["seed"]))
    assert np.uint32(weights_crc(w)) == g["weights_crc"]
    m = build_reference(cfg, w, keep_dropout=True)
    m.train()
    P_DROP, MASK_SEED = 0.1, 2468
    assert all(abs(b.dropout - P_DROP) < 1e-12 for r in m.model.per_cog_mem_bank.retrieval_blocks.values() for b in r)
    feed = MaskFeed(MASK_SEED, P_DROP)
    t = torch.from_numpy
    with inject_rng(noise=t(g["noise"]), timesteps=t(g["timesteps"]), drop_u=t(g["drop_u"])), inject_dropout(feed):
        out = m(input_ids=t(g["input_ids"]), attention_mask=t(g["attention_mask"]), images=t(g["images"]),
                actions=t(g["actions"]), indexes=[list(map(int, r)) for r in g["indexes"]])
    out.loss.backward()
    res = dict(weights_crc=g["weights_crc"], p_drop=np.float32(P_DROP), mask_seed=np.int64(MASK_SEED),
               masks_drawn=np.int64(feed.k), loss=np.float32(out.loss.item()), loss_without_dropout=g["loss"])
    gsq = 0.0
    sd = dict(m.named_parameters())
    for n, p_ in sd.items():
        if p_.grad is not None:
            gsq += float(p_.grad.double().pow(2).sum())
            res["gradN/" + n] = np.float64(p_.grad.double().norm().item())
    res["grad_norm"] = np.float64(gsq ** 0.5)
    for n in ("model.per_cog_mem_bank.retrieval_blocks.cog.1.q_proj.weight", "model.per_cog_mem_bank.retrieval_blocks.per.0.ffn.3.weight",
              "model.per_cog_mem_bank.gate_fusion_blocks.per.proj.weight"):
        res["grad/" + n] = sd[n].grad.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(GOLD, "memvla_drop_t1.npz"), **res)
    print(f"[gen_golden_memvla --dropout] {feed.k} masks, loss {res['loss']:.6f} (without dropout {float(g['loss']):.6f}) "
          f"|g| {res['grad_norm']:.4f}")


if __name__ == "__main__":
    if "--dropout" in sys.argv:
        main_dropout()
    else:
        main()
