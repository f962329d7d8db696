"""Shared test helpers: build the PRODUCT policy (dexbotic_amd) at an OracleConfig shape with the
deterministic synthetic weights of oracle/weights.py."""
import os

import numpy as np
import torch

from oracle import cogact_oracle as O
from oracle.weights import cogact_shapes, make_weights, weights_crc

CFGS = {
    "t1": O.OracleConfig(),
    "t2": O.OracleConfig(vocab_size=640, hidden_size=512, intermediate_size=768, num_hidden_layers=3,
                         num_attention_heads=4, num_key_value_heads=2, v_hidden=192, v_inter=384,
                         v_layers=4, v_heads=3, dit_hidden=192, dit_depth=3, dit_heads=3),
}


def load_golden(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, f"cogact_{tag}.npz"), allow_pickle=False)
    cfg = CFGS[tag]
    w = make_weights(cogact_shapes(cfg), int(g["seed"]))
    assert weights_crc(w) == int(g["weights_crc"])
    return g, cfg, w


def product_config(cfg: O.OracleConfig, compute_dtype="float32"):
    from dexbotic_amd.model.cogact.action_model import action_models
    from dexbotic_amd.model.cogact.action_model.dit import DiT
    from dexbotic_amd.model.cogact.cogact_arch import CogActConfig
    from dexbotic_amd.model.llm.qwen2 import Qwen2Config
    from dexbotic_amd.model.modules.mm_vision.clip.clip_encoder import CLIPVisionConfig
    action_models.DiT_models["DiT-T"] = lambda **kw: DiT(depth=cfg.dit_depth, hidden_size=cfg.dit_hidden,
                                                         num_heads=cfg.dit_heads, **kw)
    llm = Qwen2Config(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                      num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                      num_key_value_heads=cfg.num_key_value_heads, rms_norm_eps=cfg.rms_norm_eps,
                      rope_theta=cfg.rope_theta)
    vis = CLIPVisionConfig(hidden_size=cfg.v_hidden, intermediate_size=cfg.v_inter, num_hidden_layers=cfg.v_layers,
                           num_attention_heads=cfg.v_heads, image_size=cfg.v_image, patch_size=cfg.v_patch,
                           layer_norm_eps=cfg.v_eps)
    return CogActConfig(llm_config=llm, mm_vision_tower=vis, mm_projector_type="mlp2x_gelu",
                        action_model_type="DiT-T", action_dim=cfg.action_dim, chunk_size=cfg.chunk_size,
                        compute_dtype=compute_dtype)


def build_product(cfg: O.OracleConfig, weights, compute_dtype="float32", device="cuda", train=True):
    from dexbotic_amd.model.cogact.cogact_arch import CogACTForCausalLM
    m = CogACTForCausalLM(product_config(cfg, compute_dtype), device=device, train=train)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()}, strict=True)
    return m


def rel_err(a, b) -> float:
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


def assert_chunk_close(got, ref, atol: float = 1e-5, rtol: float = 1e-3, what: str = "action chunk") -> None:
    """element-wise bound on a final action chunk: |got - ref| <= atol + rtol |ref| for EVERY component (rel_err above is a
    max-norm ratio: a small component may be off by more than 1e-3 of itself and still pass it).  rtol = the tolerance
    BASELINE.json's north_star states for the predicted chunks; atol = what fp32 rounding leaves on components near zero."""
    a = np.asarray(got, dtype=np.float64)
    b = np.asarray(ref, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    excess = np.abs(a - b) - (atol + rtol * np.abs(b))
    k = int(np.argmax(excess))
    assert excess.reshape(-1)[k] <= 0, (f"{what}: component {np.unravel_index(k, a.shape)} got {a.reshape(-1)[k]!r} want "
                                        f"{b.reshape(-1)[k]!r} (atol {atol}, rtol {rtol})")


def load_lm_golden(golden_dir, tag="t1"):
    g = np.load(os.path.join(golden_dir, f"lm_{tag}.npz"), allow_pickle=False)
    cfg = CFGS[tag]
    w = {k: v for k, v in make_weights(cogact_shapes(cfg), int(g["seed"])).items() if ".action_head." not in k}
    assert weights_crc(w) == int(g["weights_crc"])
    return g, cfg, w


def build_lm_product(cfg: O.OracleConfig, weights, compute_dtype="float32", device="cuda", train=True, cls=None):
    """DexboticForCausalLM (or a subclass such as DiscreteVLAForCausalLM) at an OracleConfig shape"""
    from dexbotic_amd.model.dexbotic_arch import DexboticConfig, DexboticForCausalLM
    c = product_config(cfg, compute_dtype)
    dc = DexboticConfig(llm_config=c.llm_config, mm_vision_tower=c.mm_vision_tower, mm_projector_type="mlp2x_gelu",
                        compute_dtype=compute_dtype)
    m = (cls or DexboticForCausalLM)(dc, device=device, train=train)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()}, strict=True)
    return m
