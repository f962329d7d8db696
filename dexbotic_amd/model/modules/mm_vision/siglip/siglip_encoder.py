"""SigLIP vision tower on libdexbotic_amd kernels.

Mirror of dexbotic/model/modules/mm_vision/siglip/siglip_encoder.py:8-113 as the pi0 policy builds it
(``select_layer=None`` -> ``last_hidden_state``, dexbotic_arch.py:99-103) over the arithmetic of HF
``SiglipVisionModel`` (transformers/models/siglip/modeling_siglip.py: patch conv WITH bias, learned position
embedding, no class token, pre-LN blocks with gelu_pytorch_tanh, post_layernorm).  The attention-pooling ``head`` of
the HF module exists in checkpoints but never runs on this path; its tensors are registered (so ``state_dict`` keys
match the reference) and reported by ``unused_parameter_names``.
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
class SiglipVisionConfig:
    """subset of HF SiglipVisionConfig (defaults = google/siglip-so400m-patch14-224, the pi0 tower)"""
    hidden_size: int = 1152
    intermediate_size: int = 4304
    num_hidden_layers: int = 27
    num_attention_heads: int = 16
    image_size: int = 224
    patch_size: int = 14
    layer_norm_eps: float = 1e-6
    hidden_act: str = "gelu_pytorch_tanh"
    model_type: str = "siglip_vision_model"

    def to_dict(self):
        return asdict(self)

    @classmethod
    def from_any(cls, obj) -> "SiglipVisionConfig":
        if isinstance(obj, cls):
            return obj
        d = obj if isinstance(obj, dict) else (obj.to_dict() if hasattr(obj, "to_dict") else vars(obj))
        d = d.get("vision_config", d)
        keys = {f for f in cls.__dataclass_fields__}
        return cls(**{k: v for k, v in d.items() if k in keys and v is not None})

    @classmethod
    def from_pretrained(cls, path: str) -> "SiglipVisionConfig":
        with open(os.path.join(path, "config.json")) as f:
            return cls.from_any(json.load(f))


_ACTS = {"quick_gelu": L.ACT_QUICK_GELU, "gelu": L.ACT_GELU_ERF, "gelu_pytorch_tanh": L.ACT_GELU_TANH}


class SiglipVisionTower(nn.Module):
    def __init__(self, vision_tower_config, store: ParamStore, prefix: str = "model.mm_vision_tower.",
                 processor_config=None, delay_load=False, select_layer=None):
        super().__init__()
        if select_layer is not None:
            raise NotImplementedError("native SigLIP tower: last_hidden_state (select_layer=None), the pi0 setting")
        self.is_loaded = True
        self.select_layer = select_layer
        self.processor_config = processor_config
        self.cfg = (SiglipVisionConfig.from_pretrained(vision_tower_config) if isinstance(vision_tower_config, str)
                    else SiglipVisionConfig.from_any(vision_tower_config))
        self._image_processor = None
        self.store = store
        self.p = prefix + "vision_tower."
        c, p = self.cfg, self.p
        C_, I, P = c.hidden_size, c.intermediate_size, c.patch_size
        self.np_ = (c.image_size // P) ** 2
        self.kpad = (3 * P * P + 7) // 8 * 8
        store.new_bucket()
        store.register([(p + "embeddings.patch_embedding.weight", (C_, 3, P, P)),
                        (p + "embeddings.patch_embedding.bias", (C_,))])
        store.register([(p + "embeddings.position_embedding.weight", (self.np_, C_))])
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
        # SiglipMultiheadAttentionPoolingHead: checkpoint tensors that the VLA path never touches
        store.new_bucket()
        h = p + "head."
        store.register([(h + "probe", (1, 1, C_))])
        store.register([(h + "attention.in_proj_weight", (3 * C_, C_)), (h + "attention.in_proj_bias", (3 * C_,))])
        store.register([(h + "attention.out_proj.weight", (C_, C_)), (h + "attention.out_proj.bias", (C_,))])
        store.register([(h + "layernorm.weight", (C_,)), (h + "layernorm.bias", (C_,))], layernorm=True)
        store.register([(h + "mlp.fc1.weight", (I, C_)), (h + "mlp.fc1.bias", (I,))])
        store.register([(h + "mlp.fc2.weight", (C_, I)), (h + "mlp.fc2.bias", (C_,))])

    def unused_parameter_names(self):
        return [n for n in self.store.slots if n.startswith(self.p + "head.")]

    def load_model(self):
        return

    @property
    def image_processor(self):
        if self._image_processor is None:
            from transformers import SiglipImageProcessor    # host-side preprocessing only
            if isinstance(self.processor_config, str):
                self._image_processor = SiglipImageProcessor.from_pretrained(self.processor_config)
            else:
                s = self.cfg.image_size
                self._image_processor = SiglipImageProcessor(size={"height": s, "width": s})
            self._image_processor.crop_size = self._image_processor.size
        return self._image_processor

    def forward(self, images: torch.Tensor) -> torch.Tensor:
        """images [N,3,H,W] -> [N, N_patches, C] = post_layernorm(encoder(embeddings)) in the compute dtype"""
        if isinstance(images, list):
            images = torch.stack(images, 0)
        st, c, p = self.store, self.cfg, self.p
        N, _, H, W = images.shape
        if H != c.image_size or W != c.image_size:
            raise ValueError(f"Input image size ({H}*{W}) doesn't match model ({c.image_size}*{c.image_size}).")
        C_, P = c.hidden_size, c.patch_size
        rows = K.im2col(images.float().contiguous(), P, self.kpad, st.compute_dtype)          # [N*np, kpad]
        anchor = st.params[p + "embeddings.patch_embedding.weight"]
        patch = Fn.LinearFn.apply(rows, anchor, st, p + "embeddings.patch_embedding.weight",
                                  p + "embeddings.patch_embedding.bias", L.ACT_NONE, (C_, 3 * P * P))
        x = Fn.AddPosFn.apply(patch.view(N, self.np_, C_), anchor, st, p + "embeddings.position_embedding.weight")
        for sp in self.layer_specs:
            sp.N, sp.T = N, self.np_
            x = Fn.VitBlockFn.apply(x, st.params[sp.fc2_w], st, sp)
        x = Fn.NormFn.apply(x.reshape(N * self.np_, C_), anchor, st, "ln", p + "post_layernorm.weight",
                            p + "post_layernorm.bias", c.layer_norm_eps)
        return x.view(N, self.np_, C_)

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
