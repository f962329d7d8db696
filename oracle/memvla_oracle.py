"""CPU restatement of the reference's MemVLA policy (dexbotic/model/memvla/memvla_arch.py) — TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  MemVLA = the CogACT
path (oracle/cogact_oracle.py) + a perceptual compressor (BottleneckSE), a perceptual/cognitive memory bank with
cross-attention retrieval, gate fusion and token-merge consolidation, and a DiT whose blocks also attend to the
perceptual tokens.  Stateful exactly like the reference: samples of a batch are processed IN ORDER and each one's
(detached) fused feature is appended to its episode's bank before the next sample retrieves.

Deviation pinned in the goldens: the reference builds its retrieval blocks with dropout 0.1 and hands that value to
F.scaled_dot_product_attention unconditionally (memvla_arch.py:84-127 — active even in eval); the goldens are made
with that dropout set to 0, the only setting a deterministic comparison can pin.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch
import torch.nn.functional as F

from . import cogact_oracle as O

SD = Dict[str, torch.Tensor]
BANK = "model.per_cog_mem_bank."


def memvla_shapes(cfg: O.OracleConfig, per_token_size: int, retrieval_layers: int = 2) -> Dict[str, Tuple[int, ...]]:
    from .weights import cogact_shapes
    s = {k: v for k, v in cogact_shapes(cfg).items() if ".history_embedder." not in k}
    C, Pt = cfg.hidden_size, per_token_size
    se, hm = max(1, C // 16), max(1, int(C * 0.5))
    s["model.per_compr.excite.1.weight"] = (se, C, 1, 1)
    s["model.per_compr.excite.1.bias"] = (se,)
    s["model.per_compr.excite.3.weight"] = (C, se, 1, 1)
    s["model.per_compr.excite.3.bias"] = (C,)
    s["model.per_compr.reduce.0.weight"] = (hm, C, 1, 1)
    s["model.per_compr.reduce.0.bias"] = (hm,)
    s["model.per_compr.reduce.2.weight"] = (Pt, hm, 1, 1)
    s["model.per_compr.reduce.2.bias"] = (Pt,)
    for role, D in (("per", Pt), ("cog", C)):
        for i in range(retrieval_layers):
            bp = f"{BANK}retrieval_blocks.{role}.{i}."
            for n in ("q_proj", "k_proj", "v_proj"):
                s[bp + n + ".weight"] = (D, D)
                s[bp + n + ".bias"] = (D,)
            s[bp + "attn_norm.weight"] = (D,)
            s[bp + "attn_norm.bias"] = (D,)
            s[bp + "ffn.0.weight"] = (4 * D, D)
            s[bp + "ffn.0.bias"] = (4 * D,)
            s[bp + "ffn.3.weight"] = (D, 4 * D)
            s[bp + "ffn.3.bias"] = (D,)
            s[bp + "ffn_norm.weight"] = (D,)
            s[bp + "ffn_norm.bias"] = (D,)
    for role, D in (("per", Pt), ("cog", C)):
        s[f"{BANK}gate_fusion_blocks.{role}.proj.weight"] = (D, 2 * D)
        s[f"{BANK}gate_fusion_blocks.{role}.proj.bias"] = (D,)
    for role, D in (("per", Pt), ("cog", C)):
        s[f"{BANK}timestep_embedders.{role}.mlp.0.weight"] = (D, 256)
        s[f"{BANK}timestep_embedders.{role}.mlp.0.bias"] = (D,)
        s[f"{BANK}timestep_embedders.{role}.mlp.2.weight"] = (D, D)
        s[f"{BANK}timestep_embedders.{role}.mlp.2.bias"] = (D,)
    h = cfg.dit_hidden
    s["model.action_head.net.per_token_embedder.linear.weight"] = (h, Pt)
    s["model.action_head.net.per_token_embedder.linear.bias"] = (h,)
    for k in range(cfg.dit_depth):
        bp = f"model.action_head.net.blocks.{k}."
        s[bp + "per_attn.in_proj_weight"] = (3 * h, h)
        s[bp + "per_attn.in_proj_bias"] = (3 * h,)
        s[bp + "per_attn.out_proj.weight"] = (h, h)
        s[bp + "per_attn.out_proj.bias"] = (h,)
        s[bp + "norm3.weight"] = (h,)
        s[bp + "norm3.bias"] = (h,)
    return s


def bottleneck_se(sd: SD, x: torch.Tensor, prefix: str = "model.per_compr.") -> torch.Tensor:
    """BottleneckSE (memvla_arch.py:129-167): channel gate = sigmoid(W2 relu(W1 mean_tokens(x))), then a token-wise
    two-layer ReLU MLP down to per_token_size.  The 1x1 convolutions over the [C, H, W] view are per-token linears."""
    w = lambda n: sd[prefix + n]
    m = x.mean(dim=1)                                                            # AdaptiveAvgPool2d(1)
    g = F.relu(F.linear(m, w("excite.1.weight").flatten(1), w("excite.1.bias")))
    g = torch.sigmoid(F.linear(g, w("excite.3.weight").flatten(1), w("excite.3.bias")))
    y = x * g[:, None, :]
    y = F.relu(F.linear(y, w("reduce.0.weight").flatten(1), w("reduce.0.bias")))
    return F.linear(y, w("reduce.2.weight").flatten(1), w("reduce.2.bias"))


def cross_block(sd: SD, bp: str, query: torch.Tensor, k_in: torch.Tensor, v_in: torch.Tensor, heads: int = 4,
                mask_fn=None) -> torch.Tensor:
    """CrossTransformerBlock (memvla_arch.py:84-127): post-LN cross attention + GELU(erf) FFN.  ``mask_fn(shape)`` (training
    with the reference's dropout 0.1 left on): the three dropout masks (0 | 1/(1-p)) in the order the reference draws them —
    SDPA's dropout on the attention weights (:120-123), nn.Dropout after the GELU and after the second FFN linear (:99-105)"""
    B, N, D = query.shape
    M = k_in.shape[1]
    hd = D // heads
    q = F.linear(query, sd[bp + "q_proj.weight"], sd[bp + "q_proj.bias"]).reshape(B, N, heads, hd).transpose(1, 2)
    k = F.linear(k_in, sd[bp + "k_proj.weight"], sd[bp + "k_proj.bias"]).reshape(B, M, heads, hd).transpose(1, 2)
    v = F.linear(v_in, sd[bp + "v_proj.weight"], sd[bp + "v_proj.bias"]).reshape(B, M, heads, hd).transpose(1, 2)
    att = torch.softmax((q @ k.transpose(-1, -2)) * hd ** -0.5, dim=-1)
    draw = (lambda t_: t_ * torch.as_tensor(mask_fn(tuple(t_.shape))).to(t_.dtype)) if mask_fn is not None else (lambda t_: t_)
    att = draw(att)
    o = (att @ v).transpose(1, 2).reshape(B, N, D)
    x = F.layer_norm(query + o, (D,), sd[bp + "attn_norm.weight"], sd[bp + "attn_norm.bias"], 1e-5)
    h = draw(F.gelu(F.linear(x, sd[bp + "ffn.0.weight"], sd[bp + "ffn.0.bias"])))
    f = draw(F.linear(h, sd[bp + "ffn.3.weight"], sd[bp + "ffn.3.bias"]))
    return F.layer_norm(x + f, (D,), sd[bp + "ffn_norm.weight"], sd[bp + "ffn_norm.bias"], 1e-5)


def encode_time(sd: SD, role: str, t: torch.Tensor) -> torch.Tensor:
    """TimestepEmbedder (memvla_arch.py:36-81): sinusoid(256) -> Linear -> SiLU -> Linear"""
    p = f"{BANK}timestep_embedders.{role}."
    e = O.timestep_embedding(t, 256)
    e = F.linear(e, sd[p + "mlp.0.weight"], sd[p + "mlp.0.bias"])
    return F.linear(F.silu(e), sd[p + "mlp.2.weight"], sd[p + "mlp.2.bias"])


class MemBank:
    """PerCogMemBank state + _process_batch (memvla_arch.py:190-409), gate fusion, token-merge consolidation"""

    def __init__(self, mem_length: int, retrieval_layers: int = 2, dataloader_type: str = "group", mask_fn=None):
        self.mem_length, self.retrieval_layers, self.dataloader_type = mem_length, retrieval_layers, dataloader_type
        self.mask_fn = mask_fn                    # training with retrieval dropout: see cross_block
        self.banks = {"per": {}, "cog": {}}

    def reset(self):
        self.banks = {"per": {}, "cog": {}}

    def _consolidate(self, role, eid, feat, timestep):
        bank = self.banks[role].setdefault(eid, [])
        bank.append((timestep, feat.detach().clone()))
        while len(bank) > self.mem_length:                                  # 'tome' (memvla_arch.py:263-287)
            sims = []
            for i in range(len(bank) - 1):
                f1, f2 = bank[i][1], bank[i + 1][1]
                f1 = f1.flatten(1) if f1.dim() > 1 else f1.unsqueeze(0)
                f2 = f2.flatten(1) if f2.dim() > 1 else f2.unsqueeze(0)
                sims.append(F.cosine_similarity(f1, f2, dim=1).mean().item())
            j = int(torch.tensor(sims).argmax().item())
            (ti, fi), (tj, fj) = bank[j], bank[j + 1]
            bank[j] = (0.5 * (ti + tj), (0.5 * (fi + fj)).detach().clone())
            bank.pop(j + 1)

    def process(self, sd: SD, role: str, tokens: torch.Tensor, episode_ids, timesteps, training: bool) -> torch.Tensor:
        B, N, D = tokens.shape
        if training:
            assert self.dataloader_type == "group"
            self.banks[role].clear()
        else:
            episode_ids = [(0, 0)] * B
        outs = []
        for i in range(B):
            eid = episode_ids[i]
            working = tokens[i][None]
            hist = self.banks[role].get(eid, [])
            if hist:
                mem = torch.stack([f for _, f in hist], 0).reshape(-1, D)[None]
                ht = torch.stack([t for t, _ in hist], 0)
                pe = encode_time(sd, role, ht)[None].repeat_interleave(N, dim=1)
            else:
                mem = working
                pe = encode_time(sd, role, timesteps[i].reshape(1))[None].repeat_interleave(N, dim=1)
            q = working
            for li in range(self.retrieval_layers):
                q = cross_block(sd, f"{BANK}retrieval_blocks.{role}.{li}.", q, mem + pe, mem,
                                mask_fn=self.mask_fn if training else None)
            gp = f"{BANK}gate_fusion_blocks.{role}."
            scale = torch.sigmoid(F.linear(torch.cat([working, q], -1), sd[gp + "proj.weight"], sd[gp + "proj.bias"]))
            fused = scale * working + (1 - scale) * q
            outs.append(fused)
            self._consolidate(role, eid, fused[0], timesteps[i])
        return torch.cat(outs, 0)


def memvla_forward(sd: SD, cfg: O.OracleConfig, bank: MemBank, input_ids, attention_mask, images, actions, indexes,
                   noise, timesteps, drop_ids, repeated_diffusion_steps: int = 4, training: bool = True) -> dict:
    """MemVLAForCausalLM.forward (memvla_arch.py:546-664)"""
    feats = O.extract_vision_features(sd, cfg, images)                         # projector output = vision_proj_feats
    am = attention_mask.numpy()
    src, new_mask, _ = O.splice_plan(input_ids.numpy(), am, feats.shape[1], cfg.tokenizer_model_max_length,
                                     cfg.tokenizer_padding_side)
    hidden = O.qwen2_forward(sd, cfg, O.splice_embeds(sd, src, feats), torch.from_numpy(new_mask))
    cog = O.cognition_features(hidden, torch.from_numpy(new_mask))             # [B,1,d]
    per = bottleneck_se(sd, feats)
    eids = [tuple(ix[:2]) for ix in indexes]
    ts = [torch.tensor(ix[2]) for ix in indexes]
    cog = bank.process(sd, "cog", cog, eids, ts, training)
    per = bank.process(sd, "per", per, eids, ts, training)
    B, R = actions.shape[0], repeated_diffusion_steps
    a = actions.reshape(B, -1, cfg.action_dim)[:, :cfg.chunk_size].float().repeat(R, 1, 1)
    tab = O.training_tables(cfg.diffusion_steps)
    x_t = O.q_sample(tab, a, timesteps, noise)
    eps_hat = O.dit_forward(sd, cfg, x_t, timesteps, cog.repeat(R, 1, 1), drop_ids, per_token=per.repeat(R, 1, 1))
    loss = ((eps_hat - noise) ** 2).mean()
    return dict(loss=loss, cog=cog, per=per, eps_hat=eps_hat, hidden=hidden)


def memvla_inference_action(sd: SD, cfg: O.OracleConfig, bank: MemBank, cur_timestep: int, input_ids, images, noise,
                            action_norms, cfg_scale: float = 1.5, num_ddim_steps: int = 10):
    """MemVLAForCausalLM.inference_action (memvla_arch.py:666-746) for one frame of a running episode (the caller
    resets `bank` on the first frame and advances cur_timestep)."""
    feats = O.extract_vision_features(sd, cfg, images)
    src, _, _ = O.splice_plan(input_ids.numpy(), None, feats.shape[1], cfg.tokenizer_model_max_length,
                              cfg.tokenizer_padding_side)
    hidden = O.qwen2_forward(sd, cfg, O.splice_embeds(sd, src, feats), None)
    cog = hidden[:, -1, :][:, None, :]
    per = bottleneck_se(sd, feats)
    ts = [torch.tensor(cur_timestep)]
    cog = bank.process(sd, "cog", cog, [(0, 0)], ts, training=False)
    per = bank.process(sd, "per", per, [(0, 0)], ts, training=False)
    samples = O.ddim_sample(sd, cfg, cog, noise, cfg_scale, num_ddim_steps, per_token=per.repeat(2, 1, 1))
    return O.denorm(samples[0].numpy(), action_norms), samples
