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


# ---- full-depth fixtures (28 decoder + 24 ViT layers, 7.2 B values): the legacy generator above is one serial stream
# (~3 minutes for that many normals), so these are drawn per tensor from PCG64 streams keyed by (seed, index in sorted key
# order) — tensors are independent, can be made in parallel threads and one at a time (no 29 GB dict needed on the side that
# builds the reference).  Same numpy build on both sides (container image == GPU-box image); ``fast_weights_crc`` pins it.
def fast_weight(name: str, shape: Tuple[int, ...], seed: int, index: int, depth_scale: int = 0) -> np.ndarray:
    """one tensor of the fast family.  ``depth_scale`` = L > 0 shrinks the two residual-branch output projections of every
    decoder layer by 1/sqrt(2L) (the GPT-2 convention): with O(1) random branches a 28-layer residual stream is a chaotic map
    that amplifies ANY rounding difference, which a trained checkpoint does not do; this keeps the fixture sensitive to real
    errors instead of to noise."""
    mean, std = _scale_for(name, shape)
    if depth_scale and (name.endswith("self_attn.o_proj.weight") or name.endswith("mlp.down_proj.weight")):
        std = std / float(np.sqrt(2.0 * depth_scale))
    rng = np.random.Generator(np.random.PCG64([int(seed), int(index)]))
    out = rng.standard_normal(shape, dtype=np.float32)
    out *= np.float32(std)
    if mean:
        out += np.float32(mean)
    return out


def fast_weight_items(shapes: Dict[str, Tuple[int, ...]], seed: int, depth_scale: int = 0, threads: int = 8):
    """yields (name, fp32 array) in sorted key order, generated ``threads`` tensors ahead"""
    import concurrent.futures as cf
    names = sorted(shapes)
    with cf.ThreadPoolExecutor(max_workers=max(1, threads)) as ex:
        pending = []
        it = iter(enumerate(names))
        for _ in range(max(1, threads)):
            nx = next(it, None)
            if nx is None:
                break
            pending.append((nx[1], ex.submit(fast_weight, nx[1], shapes[nx[1]], seed, nx[0], depth_scale)))
        while pending:
            name, fut = pending.pop(0)
            arr = fut.result()
            nx = next(it, None)
            if nx is not None:
                pending.append((nx[1], ex.submit(fast_weight, nx[1], shapes[nx[1]], seed, nx[0], depth_scale)))
            yield name, arr


def fast_sample_crc(arr: np.ndarray, c: int = 0) -> int:
    """CRC32 over a strided sample of the tensor (every 4099th value + the last): cheap enough to run over 7 B values on both
    sides, still sees any change of generator, scale rule or key order"""
    flat = arr.reshape(-1)
    return zlib.crc32(np.ascontiguousarray(flat[::4099]).tobytes() + flat[-1:].tobytes(), c)
