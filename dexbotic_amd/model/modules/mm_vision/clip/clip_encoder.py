"""CLIP ViT vision tower on libdexbotic_amd kernels.

Mirror of dexbotic/model/modules/mm_vision/clip/clip_encoder.py:7-84 (``select_layer=-2``, CLS dropped,
attributes ``image_processor/hidden_size/num_patches/dtype/device/config/is_loaded``) over the
arithmetic of HF ``CLIPVisionModel`` (HF:clip/modeling_clip.py:138-218 embeddings, :259-384 encoder
layer).  Only the layers that feed ``hidden_states[-2]`` are executed: the reference computes the last
layer and post_layernorm and throws them away.
"""
from __future__ import annotations

import json
import os
from dataclasses import asdict, dataclass

import torch
import torch.nn as nn

from ..... import _lib as L
from ..... import functional as Fn
from ..... import kernels as K
from .....engine import ParamStore


@dataclass
class CLIPVisionConfig:
    """subset of HF CLIPVisionConfig (defaults = openai/clip-vit-large-patch14, the 224-px tower of the
    BASELINE config; the reference default is the 336-px variant, base_exp.py:54-56)"""
    hidden_size: int = 1024
    intermediate_size: int = 4096
    num_hidden_layers: int = 24
    num_attention_heads: int = 16
    image_size: int = 224
    patch_size: int = 14
    layer_norm_eps: float = 1e-5
    hidden_act: str = "quick_gelu"
    model_type: str = "clip_vision_model"

    def to_dict(self):
        return asdict(self)

    @classmethod
    def from_any(cls, obj) -> "CLIPVisionConfig":
        if isinstance(obj, cls):
            return obj
        d = obj if isinstance(obj, dict) else (obj.to_dict() if hasattr(obj, "to_dict") else vars(obj))
        d = d.get("vision_config", d)
        keys = {f for f in cls.__dataclass_fields__}
        return cls(**{k: v for k, v in d.items() if k in keys and v is not None})

    @classmethod
    def from_pretrained(cls, path: str) -> "CLIPVisionConfig":
        with open(os.path.join(path, "config.json")) as f:
            return cls.from_any(json.load(f))


_ACTS = {"quick_gelu": L.ACT_QUICK_GELU, "gelu": L.ACT_GELU_ERF, "gelu_pytorch_tanh": L.ACT_GELU_TANH}


class CLIPVisionTower(nn.Module):
    def __init__(self, vision_tower, store: ParamStore, prefix: str = "model.mm_vision_tower.", delay_load=False):
        super().__init__()
        self.is_loaded = True
        self.vision_tower_name = vision_tower
        self.select_layer = -2
        if isinstance(vision_tower, str):
            self.cfg = CLIPVisionConfig.from_pretrained(vision_tower)
        else:
            self.cfg = CLIPVisionConfig.from_any(vision_tower)
        self._image_processor = None
        self.store = store
        self.p = prefix + "vision_tower."
        c = self.cfg
        C_, I, P = c.hidden_size, c.intermediate_size, c.patch_size
        self.np_ = (c.image_size // P) ** 2
        self.kpad = (3 * P * P + 7) // 8 * 8            # im2col row length: 16-byte aligned rows for bf16
        p = self.p
        store.new_bucket()
        store.register([(p + "embeddings.class_embedding", (C_,))])
        store.register([(p + "embeddings.patch_embedding.weight", (C_, 3, P, P))])
        store.register([(p + "embeddings.position_embedding.weight", (self.np_ + 1, C_))])
        store.register([(p + "pre_layrnorm.weight", (C_,)), (p + "pre_layrnorm.bias", (C_,))], layernorm=True)
        self.layer_specs = []
        for j in range(c.num_hidden_layers):
            lp = f"{p}encoder.layers.{j}."
            store.new_bucket()
            qkv_w = tuple(lp + f"self_attn.{n}_proj.weight" for n in "qkv")
            qkv_b = tuple(lp + f"self_attn.{n}_proj.bias" for n in "qkv")
            store.register([(lp + "layer_norm1.weight", (C_,)), (lp + "layer_norm1.bias", (C_,))], layernorm=True)
            store.register([(n, (C_, C_)) for n in qkv_w])
            store.register([(n, (C_,)) for n in qkv_b])
            store.register([(lp + "self_attn.out_proj.weight", (C_, C_)), (lp + "self_attn.out_proj.bias", (C_,))])
            store.register([(lp + "layer_norm2.weight", (C_,)), (lp + "layer_norm2.bias", (C_,))], layernorm=True)
            store.register([(lp + "mlp.fc1.weight", (I, C_)), (lp + "mlp.fc1.bias", (I,))])
            store.register([(lp + "mlp.fc2.weight", (C_, I)), (lp + "mlp.fc2.bias", (C_,))])
            self.layer_specs.append(Fn.VitBlockSpec(
                ln1_w=lp + "layer_norm1.weight", ln1_b=lp + "layer_norm1.bias", qkv_w=qkv_w, qkv_b=qkv_b,
                out_w=lp + "self_attn.out_proj.weight", out_b=lp + "self_attn.out_proj.bias",
                ln2_w=lp + "layer_norm2.weight", ln2_b=lp + "layer_norm2.bias",
                fc1_w=lp + "mlp.fc1.weight", fc1_b=lp + "mlp.fc1.bias", fc2_w=lp + "mlp.fc2.weight",
                fc2_b=lp + "mlp.fc2.bias", act=_ACTS[c.hidden_act], eps=c.layer_norm_eps,
                H=c.num_attention_heads, D=C_ // c.num_attention_heads, I=I))
        store.new_bucket()
        store.register([(p + "post_layernorm.weight", (C_,)), (p + "post_layernorm.bias", (C_,))], layernorm=True)

    # parameters that never receive a gradient on the VLA path (last layer + post_layernorm)
    def unused_parameter_names(self):
        last = f"{self.p}encoder.layers.{self.cfg.num_hidden_layers - 1}."
        return [n for n in self.store.slots if n.startswith(last) or n.startswith(self.p + "post_layernorm.")]

    def load_model(self):
        return

    @property
    def image_processor(self):
        if self._image_processor is None:
            from transformers import CLIPImageProcessor    # host-side preprocessing only (dexbotic_arch.py:498-514)
            if isinstance(self.vision_tower_name, str):
                self._image_processor = CLIPImageProcessor.from_pretrained(self.vision_tower_name)
            else:
                s = self.cfg.image_size
                self._image_processor = CLIPImageProcessor(size={"shortest_edge": s}, crop_size={"height": s, "width": s})
        return self._image_processor

    def forward(self, images: torch.Tensor) -> torch.Tensor:
        """images [N,3,H,W] -> patch features [N, N_v, C] (= hidden_states[-2][:, 1:]) in the compute dtype."""
        if isinstance(images, list):
            images = torch.stack(images, 0)
        st, c, p = self.store, self.cfg, self.p
        N, _, H, W = images.shape
        if H != c.image_size or W != c.image_size:
            raise ValueError(f"Input image size ({H}*{W}) doesn't match model ({c.image_size}*{c.image_size}).")
        cdt = st.compute_dtype
        C_, P = c.hidden_size, c.patch_size
        rows = K.im2col(images.float().contiguous(), P, self.kpad, cdt)                 # [N*np, kpad]
        anchor = st.params[p + "embeddings.patch_embedding.weight"]
        patch = Fn.LinearFn.apply(rows, anchor, st, p + "embeddings.patch_embedding.weight", None, L.ACT_NONE,
                                  (C_, 3 * P * P))
        x = Fn.VitEmbedFn.apply(patch, anchor, st, p + "embeddings.class_embedding",
                                p + "embeddings.position_embedding.weight", N, self.np_)
        T = self.np_ + 1
        x = Fn.NormFn.apply(x.reshape(N * T, C_), anchor, st, "ln", p + "pre_layrnorm.weight", p + "pre_layrnorm.bias",
                            c.layer_norm_eps).view(N, T, C_)
        n_run = c.num_hidden_layers + 1 + self.select_layer      # hidden_states[-2] = output of layer L-1
        for sp in self.layer_specs[:n_run]:
            sp.N, sp.T = N, T
            x = Fn.VitBlockFn.apply(x, st.params[sp.fc2_w], st, sp)
        return Fn.DropClsFn.apply(x)

    @property
    def dummy_feature(self):
        return torch.zeros(1, self.hidden_size, device=self.device, dtype=self.dtype)

    @property
    def dtype(self):
        return self.store.compute_dtype

    @property
    def device(self):
        return self.store.device

    @property
    def config(self):
        return self.cfg

    @property
    def hidden_size(self):
        return self.cfg.hidden_size

    @property
    def num_patches(self):
        return self.np_
