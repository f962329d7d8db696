"""CPU oracle for the DB-CogACT hot path (TEST INFRASTRUCTURE — NOT THE PRODUCT).

This file is a plain CPU (torch fp32 / numpy) restatement of the reference's
algorithm for the path SURVEY.md §8(a) names.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it, and only as the checker.  The product path (``dexbotic_amd``) never imports
anything from ``oracle/``.

Parity pinning: the oracle is checked against golden vectors produced by
running the reference's own Python classes in the build container
(``oracle/gen_golden.py`` -> ``tests/golden/*.npz``; test:
``tests/test_oracle_golden.py``).  The reference has no tests / golden vectors
of its own, so those generated fixtures are the pin.

Every function operates on a flat ``dict[str, torch.Tensor]`` keyed exactly like
the reference ``CogACTForCausalLM.state_dict()`` and cites the
reference file:line it follows.  Paths are relative to /root/reference unless
prefixed ``HF:`` (site-packages/transformers/models, the un-vendored dependency
whose arithmetic the reference calls; pinned 4.51.0 by the reference, 5.15.0 in
this image — arithmetic-identical for these modules).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

IGNORE_INDEX = -100        # dexbotic/constants.py:1
IMAGE_TOKEN_INDEX = -200   # dexbotic/constants.py:2

SD = Dict[str, torch.Tensor]


# --------------------------------------------------------------------------- config
@dataclass
class OracleConfig:
    """Shape description of one CogACT instance (mirrors CogActConfig + sub-configs)."""
    # LLM (HF Qwen2Config)
    vocab_size: int = 512
    hidden_size: int = 256
    intermediate_size: int = 512
    num_hidden_layers: int = 2
    num_attention_heads: int = 2
    num_key_value_heads: int = 1
    rms_norm_eps: float = 1e-6
    rope_theta: float = 1e6
    # vision tower (HF CLIPVisionConfig)
    v_hidden: int = 128
    v_inter: int = 256
    v_layers: int = 3
    v_heads: int = 2
    v_image: int = 56
    v_patch: int = 14
    v_eps: float = 1e-5
    # action head (dit.py:181-243)
    dit_hidden: int = 128
    dit_depth: int = 2
    dit_heads: int = 2
    action_dim: int = 7
    chunk_size: int = 16
    diffusion_steps: int = 100
    tokenizer_model_max_length: Optional[int] = None
    tokenizer_padding_side: str = "right"

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    @property
    def num_patches(self) -> int:
        return (self.v_image // self.v_patch) ** 2


# --------------------------------------------------------------------------- vision
def clip_vision_features(sd: SD, cfg: OracleConfig, images: torch.Tensor,
                         prefix: str = "model.mm_vision_tower.vision_tower.") -> torch.Tensor:
    """CLIP ViT -> hidden_states[-2] without CLS.

    Follows dexbotic/model/modules/mm_vision/clip/clip_encoder.py:31-57
    (select_layer=-2, drop token 0) over HF:clip/modeling_clip.py:138-218 (embeddings),
    :259-384 (encoder layer: pre-LN, q/k/v/out Linear+bias, scale hd^-0.5, softmax,
    quick_gelu MLP) and pre_layrnorm.  images: [N,3,H,W] -> [N, N_v, C].
    """
    p = prefix
    C, P = cfg.v_hidden, cfg.v_patch
    N = images.shape[0]
    # Conv2d(3,C,P,P,bias=False) == unfold + matmul  (HF:clip/modeling_clip.py:149-155,206-208)
    w = sd[p + "embeddings.patch_embedding.weight"]
    patches = F.conv2d(images, w, stride=P)                      # [N,C,g,g]
    patches = patches.flatten(2).transpose(1, 2)                  # [N,N_v,C]
    cls = sd[p + "embeddings.class_embedding"].expand(N, 1, C)
    x = torch.cat([cls, patches], dim=1)
    x = x + sd[p + "embeddings.position_embedding.weight"][None]  # position_ids = arange
    x = F.layer_norm(x, (C,), sd[p + "pre_layrnorm.weight"], sd[p + "pre_layrnorm.bias"], cfg.v_eps)
    H = cfg.v_heads
    hd = C // H
    # hidden_states[-2] of a L-layer encoder = output of layer L-1 (1-based) -> run L-1 layers
    for j in range(cfg.v_layers - 1):
        lp = f"{p}encoder.layers.{j}."
        r = x
        h = F.layer_norm(x, (C,), sd[lp + "layer_norm1.weight"], sd[lp + "layer_norm1.bias"], cfg.v_eps)
        q = F.linear(h, sd[lp + "self_attn.q_proj.weight"], sd[lp + "self_attn.q_proj.bias"])
        k = F.linear(h, sd[lp + "self_attn.k_proj.weight"], sd[lp + "self_attn.k_proj.bias"])
        v = F.linear(h, sd[lp + "self_attn.v_proj.weight"], sd[lp + "self_attn.v_proj.bias"])
        T = x.shape[1]
        q = q.view(N, T, H, hd).transpose(1, 2)
        k = k.view(N, T, H, hd).transpose(1, 2)
        v = v.view(N, T, H, hd).transpose(1, 2)
        att = torch.softmax((q @ k.transpose(-1, -2)) * (hd ** -0.5), dim=-1, dtype=torch.float32)
        o = (att @ v).transpose(1, 2).reshape(N, T, C)
        o = F.linear(o, sd[lp + "self_attn.out_proj.weight"], sd[lp + "self_attn.out_proj.bias"])
        x = r + o
        r = x
        h = F.layer_norm(x, (C,), sd[lp + "layer_norm2.weight"], sd[lp + "layer_norm2.bias"], cfg.v_eps)
        h = F.linear(h, sd[lp + "mlp.fc1.weight"], sd[lp + "mlp.fc1.bias"])
        h = h * torch.sigmoid(1.702 * h)                          # quick_gelu, HF:activations.py:117-123
        h = F.linear(h, sd[lp + "mlp.fc2.weight"], sd[lp + "mlp.fc2.bias"])
        x = r + h
    return x[:, 1:]                                               # clip_encoder.py:34 drops CLS


def mm_projector(sd: SD, feats: torch.Tensor, prefix: str = "model.mm_projector.") -> torch.Tensor:
    """mlp2x_gelu projector: Linear -> GELU(erf) -> Linear.
    Follows dexbotic/model/modules/mm_projector/builder.py:71-79."""
    h = F.linear(feats, sd[prefix + "0.weight"], sd[prefix + "0.bias"])
    h = F.gelu(h)                                                 # nn.GELU() == erf form
    return F.linear(h, sd[prefix + "2.weight"], sd[prefix + "2.bias"])


def extract_vision_features(sd: SD, cfg: OracleConfig, images: torch.Tensor) -> torch.Tensor:
    """Follows dexbotic/model/dexbotic_arch.py:157-180: 5-D images are [B,V,3,H,W]; views of a
    sample are concatenated along the token axis -> [B, V*N_v, d]."""
    if images.ndim == 5:
        B, V = images.shape[:2]
        f = mm_projector(sd, clip_vision_features(sd, cfg, images.flatten(0, 1)))
        return f.reshape(B, V * f.shape[1], f.shape[2])
    return mm_projector(sd, clip_vision_features(sd, cfg, images))


# --------------------------------------------------------------------------- splice
def splice_plan(input_ids: np.ndarray, attention_mask: Optional[np.ndarray], n_img_tokens: int,
                max_length: Optional[int] = None, padding_side: str = "right"
                ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Integer plan of _prepare_inputs_labels_for_multimodal (dexbotic_arch.py:182-373).

    Returns (src, new_mask, lengths):
      src[b, s] >= 0            -> token id whose embedding sits at (b, s)
      src[b, s] == -1 - k       -> image-feature row k of sample b's image block(s)
      src[b, s] == INT_MIN      -> padding (zeros)
    Each IMAGE_TOKEN_INDEX placeholder consumes ONE image_features[cur_image_idx] block of
    n_img_tokens rows (dexbotic_arch.py:291-304); cur_image_idx is global across the batch
    (:229-233) — with one placeholder per sample it equals the sample index, the only case the
    CogACT path produces.  A sample with no placeholder still consumes one block (:264-271).
    """
    B, L = input_ids.shape
    PAD = np.iinfo(np.int64).min
    if attention_mask is None:
        attention_mask = np.ones((B, L), dtype=bool)
    attention_mask = attention_mask.astype(bool)
    rows: List[np.ndarray] = []
    img_rows: List[int] = []   # which image_features block each sample's rows come from
    cur_image_idx = 0
    for b in range(B):
        ids = input_ids[b][attention_mask[b]]                     # :219-220 strip padding
        pos = np.nonzero(ids == IMAGE_TOKEN_INDEX)[0]
        if len(pos) == 0:
            rows.append(ids.astype(np.int64))
            cur_image_idx += 1
            continue
        out: List[np.ndarray] = []
        prev = -1
        for p_ in pos:
            out.append(ids[prev + 1:p_].astype(np.int64))
            # rows of block cur_image_idx, encoded with the block index folded in
            blk = cur_image_idx
            out.append(-1 - (blk * n_img_tokens + np.arange(n_img_tokens, dtype=np.int64)))
            cur_image_idx += 1
            prev = p_
        out.append(ids[prev + 1:].astype(np.int64))
        rows.append(np.concatenate(out))
    if max_length is not None:                                    # :238-243
        rows = [r[:max_length] for r in rows]
    lengths = np.array([len(r) for r in rows], dtype=np.int64)
    S = int(lengths.max())
    src = np.full((B, S), PAD, dtype=np.int64)
    new_mask = np.zeros((B, S), dtype=bool)
    for b, r in enumerate(rows):                                  # :315-373
        n = len(r)
        if n == 0:
            continue
        if padding_side == "left":
            src[b, S - n:] = r
            new_mask[b, S - n:] = True
        else:
            src[b, :n] = r
            new_mask[b, :n] = True
    return src, new_mask, lengths


def splice_embeds(sd: SD, src: np.ndarray, image_features: torch.Tensor) -> torch.Tensor:
    """Materialise inputs_embeds from a splice plan (embed_tokens gather + image rows + zero pad).
    image_features: [num_blocks, n_img_tokens, d] (block index = global cur_image_idx)."""
    emb = sd["model.llm.embed_tokens.weight"]
    B, S = src.shape
    d = emb.shape[1]
    flat_img = image_features.reshape(-1, d).to(emb.dtype)    # torch.cat of the reference promotes bf16 (autocast) rows to fp32
    out = torch.zeros(B, S, d, dtype=emb.dtype)
    src_t = torch.from_numpy(src)
    tok = src_t >= 0
    img = (src_t < 0) & (src_t != np.iinfo(np.int64).min)
    out[tok] = emb[src_t[tok]]
    out[img] = flat_img[(-1 - src_t[img])]
    return out


# --------------------------------------------------------------------------- LLM
def rope_cos_sin(cfg: OracleConfig, position_ids: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """HF:qwen2/modeling_qwen2.py:52-104 (default rope): inv_freq = theta^(-2i/hd), emb = cat(f,f)."""
    hd = cfg.head_dim
    inv_freq = 1.0 / (cfg.rope_theta ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
    freqs = position_ids[..., None].float() * inv_freq            # [B,S,hd/2]
    emb = torch.cat([freqs, freqs], dim=-1)
    return emb.cos(), emb.sin()


def _rotate_half(x: torch.Tensor) -> torch.Tensor:
    h = x.shape[-1] // 2
    return torch.cat([-x[..., h:], x[..., :h]], dim=-1)           # HF:qwen2/modeling_qwen2.py:107-111


def rms_norm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    """HF:qwen2/modeling_qwen2.py:238-253: normalise in fp32, cast back, THEN multiply by weight."""
    x32 = x.float()
    y = x32 * torch.rsqrt(x32.pow(2).mean(-1, keepdim=True) + eps)
    return w * y.to(x.dtype)


def qwen2_forward(sd: SD, cfg: OracleConfig, inputs_embeds: torch.Tensor,
                  attention_mask: Optional[torch.Tensor], prefix: str = "model.llm.",
                  return_all: bool = False):
    """HF Qwen2Model forward on inputs_embeds (HF:qwen2/modeling_qwen2.py:258-403), called at
    dexbotic/model/cogact/cogact_arch.py:97-106.  position_ids = arange(S) for every sample
    (the reference passes position_ids=None through, dexbotic_arch.py:251); mask = causal AND
    key-padding.  Returns hidden_states[-1] (post final norm)."""
    B, S, d = inputs_embeds.shape
    H, Hkv, hd = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    pos = torch.arange(S)[None].expand(B, S)
    cos, sin = rope_cos_sin(cfg, pos)
    cos, sin = cos[:, None], sin[:, None]                         # [B,1,S,hd]
    causal = torch.tril(torch.ones(S, S, dtype=torch.bool))
    allow = causal[None, None].expand(B, 1, S, S)
    if attention_mask is not None:
        allow = allow & attention_mask.bool()[:, None, None, :]
    bias = torch.zeros(B, 1, S, S).masked_fill(~allow, float("-inf"))
    # rows with no allowed key (left-padded queries) would be NaN; HF sdpa path un-masks them
    # (fully-masked rows attend everywhere).  They are padding rows and never read.
    dead = ~allow.any(-1, keepdim=True)
    bias = bias.masked_fill(dead, 0.0)
    x = inputs_embeds
    hs = [x]
    for i in range(cfg.num_hidden_layers):
        lp = f"{prefix}layers.{i}."
        r = x
        h = rms_norm(x, sd[lp + "input_layernorm.weight"], cfg.rms_norm_eps)
        q = F.linear(h, sd[lp + "self_attn.q_proj.weight"], sd[lp + "self_attn.q_proj.bias"])
        k = F.linear(h, sd[lp + "self_attn.k_proj.weight"], sd[lp + "self_attn.k_proj.bias"])
        v = F.linear(h, sd[lp + "self_attn.v_proj.weight"], sd[lp + "self_attn.v_proj.bias"])
        q = q.view(B, S, H, hd).transpose(1, 2)
        k = k.view(B, S, Hkv, hd).transpose(1, 2)
        v = v.view(B, S, Hkv, hd).transpose(1, 2)
        q = q * cos + _rotate_half(q) * sin
        k = k * cos + _rotate_half(k) * sin
        g = H // Hkv
        k = k.repeat_interleave(g, dim=1)                         # repeat_kv
        v = v.repeat_interleave(g, dim=1)
        att = (q @ k.transpose(-1, -2)) * (hd ** -0.5) + bias
        att = torch.softmax(att, dim=-1, dtype=torch.float32).to(q.dtype)
        o = (att @ v).transpose(1, 2).reshape(B, S, H * hd)
        x = r + F.linear(o, sd[lp + "self_attn.o_proj.weight"])
        r = x
        h = rms_norm(x, sd[lp + "post_attention_layernorm.weight"], cfg.rms_norm_eps)
        gate = F.linear(h, sd[lp + "mlp.gate_proj.weight"])
        up = F.linear(h, sd[lp + "mlp.up_proj.weight"])
        x = r + F.linear(F.silu(gate) * up, sd[lp + "mlp.down_proj.weight"])
        hs.append(x)
    out = rms_norm(x, sd[prefix + "norm.weight"], cfg.rms_norm_eps)
    if return_all:
        return out, hs
    return out


def cognition_features(last_hidden: torch.Tensor, attention_mask: torch.Tensor) -> torch.Tensor:
    """dexbotic/model/cogact/cogact_arch.py:110-120: feature of the LAST un-padded token
    (first index where cumsum(mask) reaches its max)."""
    cs = attention_mask.long().cumsum(dim=1)
    idx = (cs == cs.max(dim=1, keepdim=True)[0]).float().argmax(dim=1)
    return last_hidden[torch.arange(last_hidden.shape[0]), idx][:, None, :]


# --------------------------------------------------------------------------- diffusion tables
def cosine_betas(T: int = 100, max_beta: float = 0.999) -> np.ndarray:
    """squaredcos_cap_v2: diffusion.py:205-230 (float64)."""
    ab = lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
    return np.array([min(1 - ab((i + 1) / T) / ab(i / T), max_beta) for i in range(T)], dtype=np.float64)


@dataclass
class DiffusionTables:
    """The float64 tables of GaussianDiffusion.__init__ (diffusion.py:242-292)."""
    betas: np.ndarray
    timestep_map: List[int] = field(default_factory=list)

    def __post_init__(self):
        b = self.betas.astype(np.float64)
        a = 1.0 - b
        self.alphas_cumprod = np.cumprod(a, axis=0)
        self.alphas_cumprod_prev = np.append(1.0, self.alphas_cumprod[:-1])
        self.sqrt_alphas_cumprod = np.sqrt(self.alphas_cumprod)
        self.sqrt_one_minus_alphas_cumprod = np.sqrt(1.0 - self.alphas_cumprod)
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod - 1)
        self.num_timesteps = len(b)


def _spaced(T: int, use) -> DiffusionTables:
    """SpacedDiffusion.__init__ (diffusion.py:1054-1071): betas re-derived from the retained
    cumulative products — also for the un-respaced training process (use = all steps), which is why
    the training betas differ from betas_for_alpha_bar in the last ulp."""
    base = DiffusionTables(cosine_betas(T))
    last = 1.0
    new_betas, tmap = [], []
    for i, ac in enumerate(base.alphas_cumprod):
        if i in use:
            new_betas.append(1 - ac / last)
            last = ac
            tmap.append(i)
    return DiffusionTables(np.array(new_betas), tmap)


def training_tables(T: int = 100) -> DiffusionTables:
    """create_diffusion("", "squaredcos_cap_v2", diffusion_steps=T) (action_models.py:78-83)."""
    return _spaced(T, set(range(T)))


def ddim_tables(T: int = 100, steps: int = 10) -> DiffusionTables:
    """create_diffusion("ddim<steps>") -> space_timesteps (diffusion.py:992-1017) + SpacedDiffusion."""
    use = None
    if steps == 1:
        use = {50}                                                # diffusion.py:1013-1014 (hard-coded)
    else:
        for i in range(1, T):
            if len(range(0, T, i)) == steps:
                use = set(range(0, T, i))
                break
    if use is None:
        raise ValueError(f"cannot create exactly {T} steps with an integer stride")
    return _spaced(T, use)


def q_sample(tab: DiffusionTables, x0: torch.Tensor, t: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
    """diffusion.py:308-326 with _extract_into_tensor (:975-987: float64 table -> .float())."""
    a = torch.from_numpy(tab.sqrt_alphas_cumprod)[t].float()[:, None, None]
    s = torch.from_numpy(tab.sqrt_one_minus_alphas_cumprod)[t].float()[:, None, None]
    return a * x0 + s * noise


# --------------------------------------------------------------------------- DiT
def timestep_embedding(t: torch.Tensor, dim: int = 256, max_period: float = 10000.0) -> torch.Tensor:
    """dit.py:37-57: [cos(t f) | sin(t f)], f_k = exp(-ln(max_period) k / half)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def dit_forward(sd: SD, cfg: OracleConfig, x: torch.Tensor, t: torch.Tensor, z: torch.Tensor,
                drop_ids: Optional[torch.Tensor] = None,
                prefix: str = "model.action_head.net.", per_token: Optional[torch.Tensor] = None) -> torch.Tensor:
    """DiT.forward (dit.py:273-292).  drop_ids [N] bool: LabelEmbedder.token_drop replaces z by
    `uncondition` where True (dit.py:80-96; train-mode CFG dropout, injected instead of drawn).
    Blocks: dit.py:137-162 with timm Attention(qkv_bias=True)/Mlp(GELU tanh) (un-vendored, unpinned
    `timm`; semantics restated per SURVEY.md §8c shim 1), LN no-affine eps 1e-6; FinalLayer
    dit.py:165-178; returns tokens 1..T."""
    p = prefix
    hD, Hh = cfg.dit_hidden, cfg.dit_heads
    hd = hD // Hh
    N = x.shape[0]
    xe = F.linear(x, sd[p + "x_embedder.linear.weight"], sd[p + "x_embedder.linear.bias"])
    te = timestep_embedding(t, 256)
    te = F.linear(te, sd[p + "t_embedder.mlp.0.weight"], sd[p + "t_embedder.mlp.0.bias"])
    te = F.linear(F.silu(te), sd[p + "t_embedder.mlp.2.weight"], sd[p + "t_embedder.mlp.2.bias"])
    if drop_ids is not None:
        unc = sd[p + "z_embedder.uncondition"]                   # [1, token]
        z = torch.where(drop_ids[:, None, None], unc[None].expand_as(z), z)
    ze = F.linear(z, sd[p + "z_embedder.linear.weight"], sd[p + "z_embedder.linear.bias"])
    c = te[:, None, :] + ze
    h = torch.cat([c, xe], dim=1) + sd[p + "positional_embedding"]
    T1 = h.shape[1]
    pe = None
    if per_token is not None:      # MemVLA: perceptual tokens embedded once (memvla/action_model/dit.py:315-316)
        pe = F.linear(per_token, sd[p + "per_token_embedder.linear.weight"], sd[p + "per_token_embedder.linear.bias"])
    for k in range(cfg.dit_depth):
        bp = f"{p}blocks.{k}."
        y = F.layer_norm(h, (hD,), None, None, 1e-6)
        qkv = F.linear(y, sd[bp + "attn.qkv.weight"], sd[bp + "attn.qkv.bias"])
        qkv = qkv.reshape(N, T1, 3, Hh, hd).permute(2, 0, 3, 1, 4)
        q, kk, v = qkv[0], qkv[1], qkv[2]
        att = torch.softmax((q @ kk.transpose(-1, -2)) * (hd ** -0.5), dim=-1)
        o = (att @ v).transpose(1, 2).reshape(N, T1, hD)
        h = h + F.linear(o, sd[bp + "attn.proj.weight"], sd[bp + "attn.proj.bias"])
        if pe is not None:
            # memvla/action_model/dit.py:175-185: x + MultiheadAttention(norm3(x), per, per) — nn.MultiheadAttention
            # (batch_first, bias): packed in_proj rows [q; k; v], scaled dot product, out_proj
            y = F.layer_norm(h, (hD,), sd[bp + "norm3.weight"], sd[bp + "norm3.bias"], 1e-6)
            Wi, bi = sd[bp + "per_attn.in_proj_weight"], sd[bp + "per_attn.in_proj_bias"]
            P_ = pe.shape[1]
            q = F.linear(y, Wi[:hD], bi[:hD]).reshape(N, T1, Hh, hd).transpose(1, 2)
            kk = F.linear(pe, Wi[hD:2 * hD], bi[hD:2 * hD]).reshape(N, P_, Hh, hd).transpose(1, 2)
            v = F.linear(pe, Wi[2 * hD:], bi[2 * hD:]).reshape(N, P_, Hh, hd).transpose(1, 2)
            att = torch.softmax((q @ kk.transpose(-1, -2)) * (hd ** -0.5), dim=-1)
            o = (att @ v).transpose(1, 2).reshape(N, T1, hD)
            h = h + F.linear(o, sd[bp + "per_attn.out_proj.weight"], sd[bp + "per_attn.out_proj.bias"])
        y = F.layer_norm(h, (hD,), None, None, 1e-6)
        y = F.linear(y, sd[bp + "mlp.fc1.weight"], sd[bp + "mlp.fc1.bias"])
        y = F.gelu(y, approximate="tanh")
        h = h + F.linear(y, sd[bp + "mlp.fc2.weight"], sd[bp + "mlp.fc2.bias"])
    y = F.layer_norm(h, (hD,), None, None, 1e-6)
    y = F.linear(y, sd[p + "final_layer.linear.weight"], sd[p + "final_layer.linear.bias"])
    return y[:, 1:, :]


def dit_forward_with_cfg(sd: SD, cfg: OracleConfig, x: torch.Tensor, t: torch.Tensor,
                         z: torch.Tensor, cfg_scale: float, per_token: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dit.py:294-311: both halves run on the FIRST half of x; eps = u + s (c - u)."""
    half = x[: len(x) // 2]
    out = dit_forward(sd, cfg, torch.cat([half, half], 0), t, z, per_token=per_token)
    c_eps, u_eps = torch.split(out, len(out) // 2, dim=0)
    e = u_eps + cfg_scale * (c_eps - u_eps)
    return torch.cat([e, e], dim=0)


def action_loss(sd: SD, cfg: OracleConfig, actions: torch.Tensor, cognition: torch.Tensor,
                noise: torch.Tensor, timesteps: torch.Tensor, drop_ids: Optional[torch.Tensor],
                repeated_diffusion_steps: int = 4):
    """cogact_arch.py:124-135 + ActionModel.loss (action_models.py:102-125): repeat R times,
    x_t = q_sample, eps_hat = DiT, mean squared error over all elements.  noise/timesteps/drop_ids
    are the injected random draws (shapes [R*B,T,A], [R*B], [R*B])."""
    B = actions.shape[0]
    a = actions.reshape(B, -1, cfg.action_dim)[:, :cfg.chunk_size]
    a = a.repeat(repeated_diffusion_steps, 1, 1)
    z = cognition.repeat(repeated_diffusion_steps, 1, 1)
    tab = training_tables(cfg.diffusion_steps)
    x_t = q_sample(tab, a, timesteps, noise)
    eps_hat = dit_forward(sd, cfg, x_t, timesteps, z, drop_ids)
    loss = ((eps_hat - noise) ** 2).mean()
    return loss, x_t, eps_hat


def ddim_sample(sd: SD, cfg: OracleConfig, cognition: torch.Tensor, noise: torch.Tensor,
                cfg_scale: float = 1.5, num_ddim_steps: int = 10, return_traj: bool = False,
                per_token: Optional[torch.Tensor] = None):
    """inference_action's sampler (cogact_arch.py:163-192): CFG batch [x;x], z=[cog;uncond];
    ddim_sample_loop (diffusion.py:714-794) over ddim_sample (:626-673) with eta=0,
    clip_denoised=False, FIXED_SMALL/EPSILON p_mean_variance (:351-441).  Arithmetic between
    model calls is fp32 with float64->float32 tables (:975-987)."""
    tab = ddim_tables(cfg.diffusion_steps, num_ddim_steps)
    B = cognition.shape[0]
    use_cfg = cfg_scale > 1.0
    x = noise
    z = cognition
    if use_cfg:
        x = torch.cat([noise, noise], 0)
        unc = sd["model.action_head.net.z_embedder.uncondition"][None].expand(B, 1, -1)
        z = torch.cat([cognition, unc], 0)
    f32 = lambda arr, i: torch.tensor(arr[i]).float()
    traj = []
    for i in reversed(range(tab.num_timesteps)):
        t = torch.full((x.shape[0],), tab.timestep_map[i], dtype=torch.long)   # _WrappedModel :1106-1111
        if use_cfg:
            eps = dit_forward_with_cfg(sd, cfg, x, t, z, cfg_scale, per_token=per_token)
        else:
            eps = dit_forward(sd, cfg, x, t, z, per_token=per_token)
        # _predict_xstart_from_eps (:443-448)
        x0 = f32(tab.sqrt_recip_alphas_cumprod, i) * x - f32(tab.sqrt_recipm1_alphas_cumprod, i) * eps
        # _predict_eps_from_xstart (:450-454) re-derivation
        eps2 = (f32(tab.sqrt_recip_alphas_cumprod, i) * x - x0) / f32(tab.sqrt_recipm1_alphas_cumprod, i)
        ab_prev = f32(tab.alphas_cumprod_prev, i)
        # eta = 0 -> sigma = 0 ; mean_pred = x0*sqrt(ab_prev) + sqrt(1-ab_prev-0)*eps  (:657-668)
        x = x0 * torch.sqrt(ab_prev) + torch.sqrt(1 - ab_prev - 0.0) * eps2
        traj.append(x.clone())
    if use_cfg:
        x = x[:B]
    if return_traj:
        return x, traj
    return x


# --------------------------------------------------------------------------- integer / host rows (A9)
def denorm(actions: np.ndarray, action_norms: dict) -> np.ndarray:
    """ActionOutputForCausalLM._denorm (dexbotic_arch.py:552-563)."""
    a = np.clip(actions, -1, 1)
    mn = np.array(action_norms["min"]).reshape(1, -1)
    mx = np.array(action_norms["max"]).reshape(1, -1)
    return mn + (a + 1) * 0.5 * (mx - mn)


def norm_action(action: np.ndarray, mn: np.ndarray, mx: np.ndarray) -> np.ndarray:
    """ActionNormAnd2String._norm_action (transform/action.py:378-384)."""
    mn = mn.reshape(1, -1)
    mx = mx.reshape(1, -1)
    a = np.clip(action, mn, mx)
    return (a - mn) / (mx - mn + 1e-8) * 2 - 1


def action2bin(action: np.ndarray, vocab_size: int) -> np.ndarray:
    """_action2bin (transform/action.py:386-390); numpy round = half-to-even."""
    a = np.round((action + 1) / 2 * (vocab_size - 1))
    return np.clip(a, 0, vocab_size - 1)


def bin2string(bins: np.ndarray, string_format: str = " {value}") -> List[str]:
    """_bin2string (transform/action.py:392-397)."""
    return ["".join(string_format.format(value=int(v)) for v in row) for row in bins]


def discrete_action_to_continuous(action_str: str, vocab_size: int) -> np.ndarray:
    """DiscreteVLAForCausalLM._discrete_action_to_continuous (discrete_vla_arch.py:52-58)."""
    import re
    acts = re.findall(r"\d+", action_str)[:7]
    a = np.array([int(x) for x in acts], dtype=np.float32).reshape(1, -1)
    return (a / (vocab_size - 1)) * 2 - 1


# --------------------------------------------------------------------------- optimizer (A13)
def adamw_step(params: Sequence[torch.Tensor], grads: Sequence[torch.Tensor],
               m: Sequence[torch.Tensor], v: Sequence[torch.Tensor], step: int, lr: float,
               beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8, weight_decay: float = 0.0,
               max_grad_norm: Optional[float] = 1.0) -> float:
    """torch.optim.AdamW single-tensor semantics + clip_grad_norm_ (trainer.py:25-36,88-124):
    global L2 norm over all grads, clip_coef = max_norm/(norm+1e-6) clamped to 1;
    p *= 1-lr*wd; m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
    p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps).  In-place; returns the pre-clip norm."""
    total = math.sqrt(sum(float((g.double() ** 2).sum()) for g in grads))
    coef = 1.0
    if max_grad_norm is not None:
        coef = min(1.0, max_grad_norm / (total + 1e-6))
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    for p, g, m_, v_ in zip(params, grads, m, v):
        g = g * coef
        p.mul_(1 - lr * weight_decay)
        m_.mul_(beta1).add_(g, alpha=1 - beta1)
        v_.mul_(beta2).addcmul_(g, g, value=1 - beta2)
        denom = (v_.sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(m_, denom, value=-lr / bc1)
    return total


# --------------------------------------------------------------------------- end-to-end
def cogact_forward(sd: SD, cfg: OracleConfig, input_ids: torch.Tensor,
                   attention_mask: Optional[torch.Tensor], images: torch.Tensor,
                   actions: Optional[torch.Tensor] = None, noise: Optional[torch.Tensor] = None,
                   timesteps: Optional[torch.Tensor] = None, drop_ids: Optional[torch.Tensor] = None,
                   repeated_diffusion_steps: int = 4) -> dict:
    """CogACTForCausalLM.forward (cogact_arch.py:56-147) end to end."""
    feats = extract_vision_features(sd, cfg, images)
    am = None if attention_mask is None else attention_mask.numpy()
    src, new_mask, lengths = splice_plan(input_ids.numpy(), am, feats.shape[1],
                                         cfg.tokenizer_model_max_length, cfg.tokenizer_padding_side)
    embeds = splice_embeds(sd, src, feats)
    mask_t = torch.from_numpy(new_mask)
    hidden = qwen2_forward(sd, cfg, embeds, mask_t if attention_mask is not None else None)
    out = dict(image_features=feats, inputs_embeds=embeds, attention_mask=mask_t, logits=hidden)
    if actions is not None:
        cog = cognition_features(hidden, mask_t)
        loss, x_t, eps_hat = action_loss(sd, cfg, actions.float(), cog, noise, timesteps, drop_ids,
                                         repeated_diffusion_steps)
        out.update(cognition=cog, loss=loss, x_t=x_t, eps_hat=eps_hat)
    return out


def cogact_inference_action(sd: SD, cfg: OracleConfig, input_ids: torch.Tensor, images: torch.Tensor,
                            noise: torch.Tensor, action_norms: dict, cfg_scale: float = 1.5,
                            num_ddim_steps: int = 10):
    """CogACTForCausalLM.inference_action (cogact_arch.py:149-198) with injected initial noise."""
    out = cogact_forward(sd, cfg, input_ids, None, images)
    cog = out["logits"][:, -1, :][:, None, :]
    samples, traj = ddim_sample(sd, cfg, cog, noise, cfg_scale, num_ddim_steps, return_traj=True)
    acts = denorm(samples[0].numpy(), action_norms)
    return acts, samples, traj, cog


# --------------------------------------------------------------------------- LM head / discrete decode (row A10)
IGNORE_INDEX = -100


def splice_labels(input_ids: np.ndarray, attention_mask: Optional[np.ndarray], labels: np.ndarray,
                  n_img_tokens: int, max_length: Optional[int] = None, padding_side: str = "right") -> np.ndarray:
    """labels after _prepare_inputs_labels_for_multimodal (dexbotic_arch.py:219-373): text positions keep their
    label, every image row and every padding position is IGNORE_INDEX.  Same walk as splice_plan."""
    B, L = input_ids.shape
    if attention_mask is None:
        attention_mask = np.ones((B, L), dtype=bool)
    attention_mask = attention_mask.astype(bool)
    rows: List[np.ndarray] = []
    for b in range(B):
        ids = input_ids[b][attention_mask[b]]
        lab = labels[b][attention_mask[b]]
        pos = np.nonzero(ids == IMAGE_TOKEN_INDEX)[0]
        out: List[np.ndarray] = []
        prev = -1
        for p_ in pos:
            out.append(lab[prev + 1:p_].astype(np.int64))
            out.append(np.full(n_img_tokens, IGNORE_INDEX, dtype=np.int64))
            prev = p_
        out.append(lab[prev + 1:].astype(np.int64))
        rows.append(np.concatenate(out))
    if max_length is not None:
        rows = [r[:max_length] for r in rows]
    S = max(len(r) for r in rows)
    new = np.full((B, S), IGNORE_INDEX, dtype=np.int64)
    for b, r in enumerate(rows):
        if padding_side == "left":
            new[b, S - len(r):] = r
        else:
            new[b, :len(r)] = r
    return new


def causal_lm_loss(logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """HF ForCausalLMLoss (transformers/loss/loss_utils.py, called at dexbotic_arch.py:488): logits upcast to
    fp32, labels shifted left by one (position t predicts token t+1, the last position predicts IGNORE), mean
    cross-entropy over the non-ignored positions."""
    V = logits.shape[-1]
    lg = logits.float()
    shifted = torch.nn.functional.pad(labels, (0, 1), value=IGNORE_INDEX)[..., 1:].contiguous()
    return torch.nn.functional.cross_entropy(lg.view(-1, V), shifted.view(-1), ignore_index=IGNORE_INDEX,
                                             reduction="mean")


def lm_forward(sd: SD, cfg: OracleConfig, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor],
               images: torch.Tensor, labels: Optional[torch.Tensor] = None) -> dict:
    """DexboticForCausalLM.forward (dexbotic_arch.py:429-496): VLM prefill -> lm_head logits (-> CE loss)."""
    feats = extract_vision_features(sd, cfg, images)
    am = None if attention_mask is None else attention_mask.numpy()
    src, new_mask, _ = splice_plan(input_ids.numpy(), am, feats.shape[1], cfg.tokenizer_model_max_length,
                                   cfg.tokenizer_padding_side)
    embeds = splice_embeds(sd, src, feats)
    mask_t = torch.from_numpy(new_mask)
    hidden = qwen2_forward(sd, cfg, embeds, mask_t if attention_mask is not None else None)
    logits = hidden @ sd["lm_head.weight"].t()
    out = dict(hidden=hidden, logits=logits, attention_mask=mask_t)
    if labels is not None:
        new_labels = splice_labels(input_ids.numpy(), am, labels.numpy(), feats.shape[1],
                                   cfg.tokenizer_model_max_length, cfg.tokenizer_padding_side)
        out["labels"] = torch.from_numpy(new_labels)
        out["loss"] = causal_lm_loss(logits, out["labels"])
    return out


def greedy_decode(sd: SD, cfg: OracleConfig, input_ids: torch.Tensor, images: torch.Tensor, max_new_tokens: int,
                  eos_token_id: Optional[int] = None) -> Tuple[np.ndarray, torch.Tensor]:
    """Greedy continuation (GenerationMixin.generate(do_sample=False) as DiscreteVLAForCausalLM drives it,
    discrete_vla_arch.py:33-41), restated as full-prefix recompute: append argmax(logits[:, -1]) and run the
    whole prefix again — token-for-token what a KV cache must reproduce.  Batch 1.  Returns (new ids, the
    fp32 logits each new token was chosen from)."""
    cur = input_ids.clone()
    new, rows = [], []
    for _ in range(max_new_tokens):
        out = lm_forward(sd, cfg, cur, None, images)
        row = out["logits"][0, -1].float()
        nxt = int(torch.argmax(row))
        new.append(nxt)
        rows.append(row)
        cur = torch.cat([cur, torch.tensor([[nxt]], dtype=cur.dtype)], dim=1)
        if eos_token_id is not None and nxt == eos_token_id:
            break
    return np.array(new, dtype=np.int64), torch.stack(rows)


def hybrid_forward(sd: SD, cfg: OracleConfig, input_ids, attention_mask, labels, images, actions, has_action, has_text,
                   noise, timesteps, drop_ids, repeated_diffusion_steps: int = 4) -> dict:
    """HybridCogACTForCausalLM.forward (hybrid_cogact_arch.py:59-207): one VLM prefill; text_loss = causal-LM CE
    (labels of the no-text case all ignored, then * has_text.any()); action_loss = has_action-weighted mean of the
    per-sample eps-MSE means over the R repeats, denominator sum(w) + 1e-6; loss = text_loss + action_loss."""
    feats = extract_vision_features(sd, cfg, images)
    am = attention_mask.numpy()
    src, new_mask, _ = splice_plan(input_ids.numpy(), am, feats.shape[1], cfg.tokenizer_model_max_length,
                                   cfg.tokenizer_padding_side)
    embeds = splice_embeds(sd, src, feats)
    mask_t = torch.from_numpy(new_mask)
    hidden = qwen2_forward(sd, cfg, embeds, mask_t)
    logits = hidden @ sd["lm_head.weight"].t()
    new_labels = torch.from_numpy(splice_labels(input_ids.numpy(), am, labels.numpy(), feats.shape[1],
                                                cfg.tokenizer_model_max_length, cfg.tokenizer_padding_side))
    ht = has_text.bool().view(-1)
    if not bool(ht.any()):
        new_labels[~ht] = IGNORE_INDEX
    text_loss = causal_lm_loss(logits, new_labels) * ht.any().float()
    cog = cognition_features(hidden, mask_t)
    _, x_t, eps_hat = action_loss(sd, cfg, actions.float(), cog, noise, timesteps, drop_ids, repeated_diffusion_steps)
    w = has_action.reshape(-1).float().repeat(repeated_diffusion_steps)
    per = ((eps_hat - noise) ** 2).mean(dim=[1, 2])
    a_loss = (per * w).sum() / (w.sum() + 1e-6)
    return dict(loss=text_loss + a_loss, text_loss=text_loss, action_loss=a_loss, hidden=hidden)
