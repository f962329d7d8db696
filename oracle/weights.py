"""Deterministic synthetic weights for the oracle / golden fixtures / GPU parity tests.

TEST INFRASTRUCTURE (see oracle/cogact_oracle.py header).  Weights are drawn from
``numpy.random.RandomState`` (the frozen legacy generator: bit-stable across numpy versions)
in sorted-key order, so a fixture only has to store the seed, not the tensors.

``cogact_shapes`` restates the reference's state_dict key map (verified
against the live reference by oracle/gen_golden.py, which asserts equality of keys+shapes).
"""
from __future__ import annotations

import zlib
from typing import Dict, Tuple

import numpy as np

from .cogact_oracle import OracleConfig


def cogact_shapes(cfg: OracleConfig) -> Dict[str, Tuple[int, ...]]:
    d, f, V = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size
    H, Hkv, hd = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    s: Dict[str, Tuple[int, ...]] = {}
    s["model.llm.embed_tokens.weight"] = (V, d)
    for i in range(cfg.num_hidden_layers):
        p = f"model.llm.layers.{i}."
        s[p + "self_attn.q_proj.weight"] = (H * hd, d)
        s[p + "self_attn.q_proj.bias"] = (H * hd,)
        s[p + "self_attn.k_proj.weight"] = (Hkv * hd, d)
        s[p + "self_attn.k_proj.bias"] = (Hkv * hd,)
        s[p + "self_attn.v_proj.weight"] = (Hkv * hd, d)
        s[p + "self_attn.v_proj.bias"] = (Hkv * hd,)
        s[p + "self_attn.o_proj.weight"] = (d, H * hd)
        s[p + "mlp.gate_proj.weight"] = (f, d)
        s[p + "mlp.up_proj.weight"] = (f, d)
        s[p + "mlp.down_proj.weight"] = (d, f)
        s[p + "input_layernorm.weight"] = (d,)
        s[p + "post_attention_layernorm.weight"] = (d,)
    s["model.llm.norm.weight"] = (d,)
    C, I, P = cfg.v_hidden, cfg.v_inter, cfg.v_patch
    v = "model.mm_vision_tower.vision_tower."  # transformers>=5 key layout (4.51 had an extra ".vision_model")
    s[v + "embeddings.class_embedding"] = (C,)
    s[v + "embeddings.patch_embedding.weight"] = (C, 3, P, P)
    s[v + "embeddings.position_embedding.weight"] = (cfg.num_patches + 1, C)
    s[v + "pre_layrnorm.weight"] = (C,)
    s[v + "pre_layrnorm.bias"] = (C,)
    for j in range(cfg.v_layers):
        p = f"{v}encoder.layers.{j}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            s[p + f"self_attn.{n}.weight"] = (C, C)
            s[p + f"self_attn.{n}.bias"] = (C,)
        for n in ("layer_norm1", "layer_norm2"):
            s[p + n + ".weight"] = (C,)
            s[p + n + ".bias"] = (C,)
        s[p + "mlp.fc1.weight"] = (I, C)
        s[p + "mlp.fc1.bias"] = (I,)
        s[p + "mlp.fc2.weight"] = (C, I)
        s[p + "mlp.fc2.bias"] = (C,)
    s[v + "post_layernorm.weight"] = (C,)
    s[v + "post_layernorm.bias"] = (C,)
    s["model.mm_projector.0.weight"] = (d, C)
    s["model.mm_projector.0.bias"] = (d,)
    s["model.mm_projector.2.weight"] = (d, d)
    s["model.mm_projector.2.bias"] = (d,)
    a = "model.action_head.net."
    h, A, T = cfg.dit_hidden, cfg.action_dim, cfg.chunk_size
    s[a + "positional_embedding"] = (T + 1, h)
    s[a + "history_embedder.linear.weight"] = (h, A)
    s[a + "history_embedder.linear.bias"] = (h,)
    s[a + "x_embedder.linear.weight"] = (h, A)
    s[a + "x_embedder.linear.bias"] = (h,)
    s[a + "t_embedder.mlp.0.weight"] = (h, 256)
    s[a + "t_embedder.mlp.0.bias"] = (h,)
    s[a + "t_embedder.mlp.2.weight"] = (h, h)
    s[a + "t_embedder.mlp.2.bias"] = (h,)
    s[a + "z_embedder.uncondition"] = (1, d)
    s[a + "z_embedder.linear.weight"] = (h, d)
    s[a + "z_embedder.linear.bias"] = (h,)
    for k in range(cfg.dit_depth):
        p = f"{a}blocks.{k}."
        s[p + "attn.qkv.weight"] = (3 * h, h)
        s[p + "attn.qkv.bias"] = (3 * h,)
        s[p + "attn.proj.weight"] = (h, h)
        s[p + "attn.proj.bias"] = (h,)
        s[p + "mlp.fc1.weight"] = (4 * h, h)
        s[p + "mlp.fc1.bias"] = (4 * h,)
        s[p + "mlp.fc2.weight"] = (h, 4 * h)
        s[p + "mlp.fc2.bias"] = (h,)
    s[a + "final_layer.linear.weight"] = (A, h)
    s[a + "final_layer.linear.bias"] = (A,)
    s["lm_head.weight"] = (V, d)
    return s


def _scale_for(name: str, shape: Tuple[int, ...]) -> Tuple[float, float]:
    """(mean, std) heuristics that keep activations O(1) through every block."""
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "bias":
        return 0.0, 0.05
    is_norm = any(t in name for t in ("layernorm", "layer_norm", "layrnorm", ".norm."))
    if is_norm and leaf == "weight":
        return 1.0, 0.1
    if leaf in ("class_embedding", "positional_embedding", "uncondition") or \
            "position_embedding" in name:
        return 0.0, 0.3
    if "embed_tokens" in name:
        return 0.0, 0.5
    fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
    return 0.0, 1.0 / np.sqrt(fan_in)


def make_weights(shapes: Dict[str, Tuple[int, ...]], seed: int) -> Dict[str, np.ndarray]:
    rs = np.random.RandomState(seed)
    out = {}
    for name in sorted(shapes):
        mean, std = _scale_for(name, shapes[name])
        out[name] = (mean + std * rs.standard_normal(shapes[name])).astype(np.float32)
    return out


def weights_crc(w: Dict[str, np.ndarray]) -> int:
    """CRC32 over all tensors in sorted order — stored in fixtures to prove regeneration is exact."""
    c = 0
    for name in sorted(w):
        c = zlib.crc32(np.ascontiguousarray(w[name]).tobytes(), c)
    return c
