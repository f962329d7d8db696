"""CPU restatement of the reference's pi0 policy (dexbotic/model/pi0/pi0_arch.py) — TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path
(dexbotic_amd/) never does.  torch fp32 on CPU, each function citing the reference lines it follows; third-party
arithmetic (HF Gemma / SigLIP modelling code, transformers 4.51 semantics with the single sqrt(d) embedding scale,
SURVEY.md §8c shim 2-i) is restated from the published module code.  Pinned against golden vectors produced by the
live reference (oracle/gen_golden_pi0.py -> tests/golden/pi0_t1.npz).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


@dataclass
class Pi0OracleConfig:
    # Gemma "llm" expert
    vocab_size: int = 300
    hidden_size: int = 128
    intermediate_size: int = 192
    num_hidden_layers: int = 2
    num_attention_heads: int = 2
    num_key_value_heads: int = 1
    head_dim: int = 32
    rope_theta: float = 10000.0
    rms_norm_eps: float = 1e-6
    # Gemma "action expert" (same depth / heads / head_dim — the two experts share one attention)
    a_hidden: int = 64
    a_inter: int = 96
    # SigLIP vision tower
    v_hidden: int = 64
    v_inter: int = 96
    v_layers: int = 2
    v_heads: int = 2
    v_image: int = 28
    v_patch: int = 14
    v_eps: float = 1e-6
    # policy
    action_dim: int = 32
    chunk_size: int = 6

    @property
    def num_patches(self) -> int:
        return (self.v_image // self.v_patch) ** 2


def pi0_shapes(c: Pi0OracleConfig) -> Dict[str, Tuple[int, ...]]:
    """state_dict key -> shape of Pi0ForCausalLM (asserted against the live reference by gen_golden_pi0.py)"""
    s: Dict[str, Tuple[int, ...]] = {}

    def gemma(p, d, f):
        s[p + "embed_tokens.weight"] = (c.vocab_size, d)
        for i in range(c.num_hidden_layers):
            lp = f"{p}layers.{i}."
            s[lp + "self_attn.q_proj.weight"] = (c.num_attention_heads * c.head_dim, d)
            s[lp + "self_attn.k_proj.weight"] = (c.num_key_value_heads * c.head_dim, d)
            s[lp + "self_attn.v_proj.weight"] = (c.num_key_value_heads * c.head_dim, d)
            s[lp + "self_attn.o_proj.weight"] = (d, c.num_attention_heads * c.head_dim)
            s[lp + "mlp.gate_proj.weight"] = (f, d)
            s[lp + "mlp.up_proj.weight"] = (f, d)
            s[lp + "mlp.down_proj.weight"] = (d, f)
            s[lp + "input_layernorm.weight"] = (d,)
            s[lp + "post_attention_layernorm.weight"] = (d,)
        s[p + "norm.weight"] = (d,)

    gemma("model.llm.", c.hidden_size, c.intermediate_size)
    v = "model.mm_vision_tower.vision_tower."
    dv = c.v_hidden
    s[v + "embeddings.patch_embedding.weight"] = (dv, 3, c.v_patch, c.v_patch)
    s[v + "embeddings.patch_embedding.bias"] = (dv,)
    s[v + "embeddings.position_embedding.weight"] = (c.num_patches, dv)
    for i in range(c.v_layers):
        lp = f"{v}encoder.layers.{i}."
        for n in ("layer_norm1", "layer_norm2"):
            s[lp + n + ".weight"] = (dv,)
            s[lp + n + ".bias"] = (dv,)
        for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
            s[lp + f"self_attn.{n}.weight"] = (dv, dv)
            s[lp + f"self_attn.{n}.bias"] = (dv,)
        s[lp + "mlp.fc1.weight"] = (c.v_inter, dv)
        s[lp + "mlp.fc1.bias"] = (c.v_inter,)
        s[lp + "mlp.fc2.weight"] = (dv, c.v_inter)
        s[lp + "mlp.fc2.bias"] = (dv,)
    s[v + "post_layernorm.weight"] = (dv,)
    s[v + "post_layernorm.bias"] = (dv,)
    # SiglipMultiheadAttentionPoolingHead: present in the state_dict, never used (select_layer=None reads
    # last_hidden_state, siglip_encoder.py:60-64)
    s[v + "head.probe"] = (1, 1, dv)
    s[v + "head.attention.in_proj_weight"] = (3 * dv, dv)
    s[v + "head.attention.in_proj_bias"] = (3 * dv,)
    s[v + "head.attention.out_proj.weight"] = (dv, dv)
    s[v + "head.attention.out_proj.bias"] = (dv,)
    s[v + "head.layernorm.weight"] = (dv,)
    s[v + "head.layernorm.bias"] = (dv,)
    s[v + "head.mlp.fc1.weight"] = (c.v_inter, dv)
    s[v + "head.mlp.fc1.bias"] = (c.v_inter,)
    s[v + "head.mlp.fc2.weight"] = (dv, c.v_inter)
    s[v + "head.mlp.fc2.bias"] = (dv,)
    s["model.mm_projector.weight"] = (c.hidden_size, dv)
    s["model.mm_projector.bias"] = (c.hidden_size,)
    gemma("model.action_expert.", c.a_hidden, c.a_inter)
    da = c.a_hidden
    s["model.state_proj.weight"] = (da, c.action_dim)
    s["model.state_proj.bias"] = (da,)
    s["model.action_in_proj.weight"] = (da, c.action_dim)
    s["model.action_in_proj.bias"] = (da,)
    s["model.action_time_mlp_in.weight"] = (da, 2 * da)
    s["model.action_time_mlp_in.bias"] = (da,)
    s["model.action_time_mlp_out.weight"] = (da, da)
    s["model.action_time_mlp_out.bias"] = (da,)
    s["model.action_out_proj.weight"] = (c.action_dim, da)
    s["model.action_out_proj.bias"] = (c.action_dim,)
    return s


# ----------------------------------------------------------------------------------------- SigLIP tower
def siglip_features(sd: SD, c: Pi0OracleConfig, images: torch.Tensor) -> torch.Tensor:
    """SiglipVisionModel(images).last_hidden_state (select_layer=None, siglip_encoder.py:60-83): patch conv with
    bias + learned position embedding (no class token), pre-LN blocks with gelu_pytorch_tanh MLP, post_layernorm.
    HF transformers/models/siglip/modeling_siglip.py: SiglipVisionEmbeddings, SiglipEncoderLayer,
    SiglipVisionTransformer."""
    v = "model.mm_vision_tower.vision_tower."
    x = F.conv2d(images, sd[v + "embeddings.patch_embedding.weight"], sd[v + "embeddings.patch_embedding.bias"],
                 stride=c.v_patch)
    x = x.flatten(2).transpose(1, 2) + sd[v + "embeddings.position_embedding.weight"][None]
    H, hd = c.v_heads, c.v_hidden // c.v_heads
    for i in range(c.v_layers):
        lp = f"{v}encoder.layers.{i}."
        h = F.layer_norm(x, (c.v_hidden,), sd[lp + "layer_norm1.weight"], sd[lp + "layer_norm1.bias"], c.v_eps)
        B, N, _ = h.shape
        q = F.linear(h, sd[lp + "self_attn.q_proj.weight"], sd[lp + "self_attn.q_proj.bias"]).view(B, N, H, hd).transpose(1, 2)
        k = F.linear(h, sd[lp + "self_attn.k_proj.weight"], sd[lp + "self_attn.k_proj.bias"]).view(B, N, H, hd).transpose(1, 2)
        vv = F.linear(h, sd[lp + "self_attn.v_proj.weight"], sd[lp + "self_attn.v_proj.bias"]).view(B, N, H, hd).transpose(1, 2)
        w = torch.softmax((q @ k.transpose(-1, -2)) * hd ** -0.5, dim=-1, dtype=torch.float32).to(q.dtype)
        o = (w @ vv).transpose(1, 2).reshape(B, N, c.v_hidden)
        x = x + F.linear(o, sd[lp + "self_attn.out_proj.weight"], sd[lp + "self_attn.out_proj.bias"])
        h = F.layer_norm(x, (c.v_hidden,), sd[lp + "layer_norm2.weight"], sd[lp + "layer_norm2.bias"], c.v_eps)
        h = F.linear(h, sd[lp + "mlp.fc1.weight"], sd[lp + "mlp.fc1.bias"])
        h = F.gelu(h, approximate="tanh")
        x = x + F.linear(h, sd[lp + "mlp.fc2.weight"], sd[lp + "mlp.fc2.bias"])
    return F.layer_norm(x, (c.v_hidden,), sd[v + "post_layernorm.weight"], sd[v + "post_layernorm.bias"], c.v_eps)


def encode_images(sd: SD, c: Pi0OracleConfig, images: torch.Tensor) -> torch.Tensor:
    """pi0_arch.py:218-221: vision tower then the linear projector"""
    return F.linear(siglip_features(sd, c, images), sd["model.mm_projector.weight"], sd["model.mm_projector.bias"])


# ------------------------------------------------------------------------------------------- Gemma pieces
def gemma_rms_norm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    """GemmaRMSNorm: fp32 normalise, scale by (1 + weight), cast back (HF gemma/modeling_gemma.py)"""
    xf = x.float()
    out = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)
    return (out * (1.0 + w.float())).type_as(x)


def rope_cos_sin(c: Pi0OracleConfig, positions: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """GemmaRotaryEmbedding(x, position_ids) -> cos, sin [B, S, head_dim]"""
    inv = 1.0 / (c.rope_theta ** (torch.arange(0, c.head_dim, 2, dtype=torch.float32) / c.head_dim))
    fr = positions[:, :, None].float() * inv[None, None, :]
    emb = torch.cat([fr, fr], dim=-1)
    return emb.cos(), emb.sin()


def _rot_half(x):
    h = x.shape[-1] // 2
    return torch.cat([-x[..., h:], x[..., :h]], dim=-1)


def make_attn_mask(input_mask: torch.Tensor, ar_mask: torch.Tensor) -> torch.Tensor:
    """pi0_arch.py:22-28: key j visible to query i iff cumsum(ar)[j] <= cumsum(ar)[i] and both are valid"""
    ar = ar_mask.broadcast_to(input_mask.shape)
    cum = torch.cumsum(ar, dim=1)
    return (cum[:, None, :] <= cum[:, :, None]) & (input_mask[:, None, :] & input_mask[:, :, None])


NEG = -2.3819763e38          # pi0_arch.py:31-33


def inner_forward_mot(sd: SD, c: Pi0OracleConfig, embeds: List[Optional[torch.Tensor]], mask_bool: torch.Tensor,
                      cos: torch.Tensor, sin: torch.Tensor, past: Optional[List[Tuple[torch.Tensor, torch.Tensor]]] = None,
                      collect_cache: bool = False):
    """pi0_arch.py:116-216: both experts' Q/K/V of a layer are concatenated on the sequence axis, ONE attention
    (eager: additive mask, fp32 softmax), outputs split back to each expert's o_proj + GeGLU MLP; final per-expert
    norm.  `past` = per-layer (K, V) of a cached prefix that is PREPENDED without being updated (:177-183)."""
    prefixes = ["model.llm.", "model.action_expert."]
    H, Hkv, hd = c.num_attention_heads, c.num_key_value_heads, c.head_dim
    add_mask = torch.where(mask_bool, 0.0, NEG)[:, None]
    cache = []
    xs = list(embeds)
    for li in range(c.num_hidden_layers):
        qs, ks, vs, lens = [], [], [], []
        for p, x in zip(prefixes, xs):
            if x is None:
                lens.append(0)
                continue
            lp = f"{p}layers.{li}."
            h = gemma_rms_norm(x, sd[lp + "input_layernorm.weight"], c.rms_norm_eps)
            B, S, _ = h.shape
            lens.append(S)
            qs.append(F.linear(h, sd[lp + "self_attn.q_proj.weight"]).view(B, S, H, hd).transpose(1, 2))
            ks.append(F.linear(h, sd[lp + "self_attn.k_proj.weight"]).view(B, S, Hkv, hd).transpose(1, 2))
            vs.append(F.linear(h, sd[lp + "self_attn.v_proj.weight"]).view(B, S, Hkv, hd).transpose(1, 2))
        q, k, v = torch.cat(qs, 2), torch.cat(ks, 2), torch.cat(vs, 2)
        q = q * cos[:, None] + _rot_half(q) * sin[:, None]
        k = k * cos[:, None] + _rot_half(k) * sin[:, None]
        if collect_cache:
            cache.append((k, v))
        if past is not None:
            k = torch.cat([past[li][0], k], dim=-2)
            v = torch.cat([past[li][1], v], dim=-2)
        kk = k.repeat_interleave(H // Hkv, dim=1)
        vv = v.repeat_interleave(H // Hkv, dim=1)
        w = (q @ kk.transpose(-1, -2)) * hd ** -0.5 + add_mask
        w = torch.softmax(w, dim=-1, dtype=torch.float32).to(q.dtype)
        o = (w @ vv).transpose(1, 2).reshape(q.shape[0], sum(lens), H * hd)
        start, nxt = 0, []
        for p, x, n in zip(prefixes, xs, lens):
            if n == 0:
                nxt.append(None)
                continue
            lp = f"{p}layers.{li}."
            a = F.linear(o[:, start:start + n], sd[lp + "self_attn.o_proj.weight"])
            start += n
            r = x + a
            h = gemma_rms_norm(r, sd[lp + "post_attention_layernorm.weight"], c.rms_norm_eps)
            g = F.gelu(F.linear(h, sd[lp + "mlp.gate_proj.weight"]), approximate="tanh") * F.linear(h, sd[lp + "mlp.up_proj.weight"])
            nxt.append(r + F.linear(g, sd[lp + "mlp.down_proj.weight"]))
        xs = nxt
    outs = [None if x is None else gemma_rms_norm(x, sd[p + "norm.weight"], c.rms_norm_eps) for p, x in zip(prefixes, xs)]
    return outs, cache


def embed_prefix(sd: SD, c: Pi0OracleConfig, input_ids, attention_mask, images, image_masks):
    """pi0_arch.py:223-259: per-camera image tokens (mask broadcast over the camera's tokens), then text tokens
    scaled by sqrt(hidden); all prefix tokens are one bidirectional block (ar_mask False)."""
    toks, masks = [], []
    for cam in range(images.shape[1]):
        t = encode_images(sd, c, images[:, cam])
        toks.append(t)
        masks.append(image_masks[:, cam, None].expand(t.shape[0], t.shape[1]))
    toks.append(sd["model.llm.embed_tokens.weight"][input_ids] * c.hidden_size ** 0.5)
    masks.append(attention_mask.bool())
    tokens = torch.cat(toks, 1)
    return tokens, torch.cat(masks, 1), torch.zeros(tokens.shape[1], dtype=torch.bool)


def posemb_sincos(t: torch.Tensor, dim: int, min_period: float = 4e-3, max_period: float = 4.0) -> torch.Tensor:
    """pi0_arch.py:36-51 (fraction in float64, the division lands in float64, the result is float64)"""
    frac = torch.linspace(0.0, 1.0, dim // 2, dtype=torch.float64)
    period = min_period * (max_period / min_period) ** frac
    x = t[:, None].float() / period[None, :] * 2 * np.pi
    return torch.cat([torch.sin(x), torch.cos(x)], dim=-1)


def embed_suffix(sd: SD, c: Pi0OracleConfig, states, noisy_actions, time):
    """pi0_arch.py:261-315: state token (its own block), then chunk action tokens = MLP([action_proj ; sincos(t)])
    (one block starting at the first action token)."""
    st = F.linear(states, sd["model.state_proj.weight"], sd["model.state_proj.bias"])[:, None]
    te = posemb_sincos(time, c.a_hidden)[:, None].expand(-1, c.chunk_size, -1)
    at = F.linear(noisy_actions, sd["model.action_in_proj.weight"], sd["model.action_in_proj.bias"])
    h = torch.cat([at, te.to(at.dtype)], dim=-1)
    h = F.linear(h, sd["model.action_time_mlp_in.weight"], sd["model.action_time_mlp_in.bias"])
    h = F.silu(h)
    h = F.linear(h, sd["model.action_time_mlp_out.weight"], sd["model.action_time_mlp_out.bias"])
    tokens = torch.cat([st, h], 1)
    mask = torch.ones(tokens.shape[:2], dtype=torch.bool)
    ar = torch.tensor([True, True] + [False] * (c.chunk_size - 1))
    return tokens, mask, ar


def pi0_forward(sd: SD, c: Pi0OracleConfig, input_ids, attention_mask, images, image_masks, states, actions,
                noise, time) -> dict:
    """Pi0ForCausalLM.forward (pi0_arch.py:317-400) with the random draws (noise ~ N(0,1), time ~ Beta(1.5,1)
    *0.999+0.001) injected: flow-matching target u = noise - actions at x_t = t noise + (1-t) actions."""
    te = time[:, None, None]
    x_t = te * noise + (1 - te) * actions
    u_t = noise - actions
    ptok, pmask, par = embed_prefix(sd, c, input_ids, attention_mask, images, image_masks)
    stok, smask, sar = embed_suffix(sd, c, states, x_t, time)
    input_mask = torch.cat([pmask, smask], 1)
    ar = torch.cat([par, sar], 0)
    mask = make_attn_mask(input_mask, ar)
    positions = torch.cumsum(input_mask, dim=1) - 1
    cos, sin = rope_cos_sin(c, positions)
    (pre, suf), _ = inner_forward_mot(sd, c, [ptok, stok], mask, cos, sin)
    v_t = F.linear(suf[:, -c.chunk_size:], sd["model.action_out_proj.weight"], sd["model.action_out_proj.bias"])
    loss = F.mse_loss(v_t, u_t, reduction="none").mean()
    return dict(prefix_tokens=ptok, suffix_tokens=stok, prefix_out=pre, suffix_out=suf, v_t=v_t, loss=loss, x_t=x_t)


def pi0_inference_action(sd: SD, c: Pi0OracleConfig, input_ids, attention_mask, states, images, image_masks,
                         noise: torch.Tensor, diffusion_steps: int = 10) -> torch.Tensor:
    """Pi0ForCausalLM.inference_action (pi0_arch.py:402-491): prefix pass fills the K/V cache; `diffusion_steps`
    Euler steps x += v dt from t = 1 to 0, each re-encoding the suffix against the cached prefix."""
    B = states.shape[0]
    dt = -1.0 / diffusion_steps
    ptok, pmask, par = embed_prefix(sd, c, input_ids, attention_mask, images, image_masks)
    pm = make_attn_mask(pmask, par)
    positions = torch.cumsum(pmask, dim=1) - 1
    cos, sin = rope_cos_sin(c, positions)
    _, cache = inner_forward_mot(sd, c, [ptok, None], pm, cos, sin, collect_cache=True)
    x, time = noise, torch.tensor(1.0)
    while time > -dt / 2:
        stok, smask, sar = embed_suffix(sd, c, states, x, time.broadcast_to(B))
        sm = make_attn_mask(smask, sar)
        full = torch.cat([pmask[:, None, :].repeat(1, stok.shape[1], 1), sm], dim=-1)
        fpos = pmask.sum(-1)[:, None] + torch.cumsum(smask, dim=-1) - 1
        cos, sin = rope_cos_sin(c, fpos)
        (_, suf), _ = inner_forward_mot(sd, c, [None, stok], full, cos, sin, past=cache)
        v_t = F.linear(suf[:, -c.chunk_size:], sd["model.action_out_proj.weight"], sd["model.action_out_proj.bias"])
        x, time = x + v_t * dt, time + dt
    return x
