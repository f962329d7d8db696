"""Gemma decoder expert on libdexbotic_amd kernels (the two backbones of the pi0 mixture of transformers).

Stands in for the HF ``GemmaModel`` objects the reference builds through ``AutoModel.from_config`` (pi0_arch.py:
86-91) and walks layer by layer in ``_inner_forward_mot`` (:116-216).  Arithmetic per HF gemma/modeling_gemma.py:
GemmaRMSNorm (fp32 normalise, scale by 1 + weight), bias-free q/k/v/o, rotate-half RoPE, GeGLU MLP
(gelu_pytorch_tanh(gate) * up), final norm.  Parameter names are HF's.  A layer is split in the two halves the
mixture needs: ``pre_attention`` (norm + fused QKV) and ``post_attention`` (o_proj + residual + norm + MLP +
residual); the attention itself runs once over BOTH experts' tokens (pi0_arch.py:161-191).
"""
from __future__ import annotations

from dataclasses import asdict, dataclass
from typing import Dict, Tuple

import torch
import torch.nn as nn

from ... import _lib as L
from ... import functional as Fn
from ... import kernels as K
from ...engine import ParamStore


@dataclass
class GemmaConfig:
    """subset of HF GemmaConfig that defines the arithmetic (defaults = the 2 B backbone of pi0)"""
    vocab_size: int = 257152
    hidden_size: int = 2048
    intermediate_size: int = 16384
    num_hidden_layers: int = 18
    num_attention_heads: int = 8
    num_key_value_heads: int = 1
    head_dim: int = 256
    rms_norm_eps: float = 1e-6
    rope_theta: float = 10000.0
    max_position_embeddings: int = 8192
    model_type: str = "gemma"

    def to_dict(self):
        return asdict(self)

    @classmethod
    def from_any(cls, obj) -> "GemmaConfig":
        if isinstance(obj, cls):
            return obj
        d = obj if isinstance(obj, dict) else (obj.to_dict() if hasattr(obj, "to_dict") else vars(obj))
        rp = d.get("rope_parameters") or {}
        keys = {f for f in cls.__dataclass_fields__}
        kw = {k: v for k, v in d.items() if k in keys and v is not None}
        kw["rope_theta"] = d.get("rope_theta", rp.get("rope_theta", 10000.0))
        if kw.get("model_type", "gemma") != "gemma":
            raise NotImplementedError(f"pi0 experts are Gemma models, got model_type={kw['model_type']!r}")
        return cls(**kw)


class GemmaExpert(nn.Module):
    def __init__(self, store: ParamStore, prefix: str, config: GemmaConfig):
        super().__init__()
        self.store, self.p, self.config = store, prefix, config
        c = config
        d, f, hd, Hq, Hkv = c.hidden_size, c.intermediate_size, c.head_dim, c.num_attention_heads, c.num_key_value_heads
        store.new_bucket()
        store.register([(prefix + "embed_tokens.weight", (c.vocab_size, d))])
        self.layer_names = []
        self.layer_specs = []
        for i in range(c.num_hidden_layers):
            lp = f"{prefix}layers.{i}."
            store.new_bucket()
            qkv = tuple(lp + f"self_attn.{n}_proj.weight" for n in "qkv")
            gu = (lp + "mlp.gate_proj.weight", lp + "mlp.up_proj.weight")
            store.register([(lp + "input_layernorm.weight", (d,))])
            store.register([(qkv[0], (Hq * hd, d)), (qkv[1], (Hkv * hd, d)), (qkv[2], (Hkv * hd, d))])
            store.register([(lp + "self_attn.o_proj.weight", (d, Hq * hd))])
            store.register([(lp + "post_attention_layernorm.weight", (d,))])
            store.register([(gu[0], (f, d)), (gu[1], (f, d))])
            store.register([(lp + "mlp.down_proj.weight", (d, f))])
            self.layer_names.append(dict(ln1=lp + "input_layernorm.weight", qkv=qkv, o=lp + "self_attn.o_proj.weight",
                                    ln2=lp + "post_attention_layernorm.weight", gu=gu, down=lp + "mlp.down_proj.weight"))
            self.layer_specs.append(Fn.GemmaLayerSpec(d=d, F=f, eps=c.rms_norm_eps, **self.layer_names[-1]))
        store.new_bucket()
        store.register([(prefix + "norm.weight", (d,))])
        self._rope: Dict = {}
        self._w1: Dict[str, torch.Tensor] = {}

    @property
    def embed_name(self) -> str:
        return self.p + "embed_tokens.weight"

    # ---- pieces -------------------------------------------------------------------------------------
    def norm_weight(self, name: str) -> torch.Tensor:
        """1 + weight in fp32 (GemmaRMSNorm multiplies the normalised fp32 activations by (1 + w.float()));
        cached while the module is in eval mode, rebuilt every call in training (the weights move)."""
        if not self.training and name in self._w1:
            return self._w1[name]
        w1 = self.store.w(name).float() + 1.0
        if not self.training:
            self._w1[name] = w1
        return w1

    def train(self, mode: bool = True):
        self._w1.clear()
        return super().train(mode)

    def rope_tables(self, n_pos: int, device) -> Tuple[torch.Tensor, torch.Tensor]:
        """cos/sin [n_pos, head_dim/2] fp32 (GemmaRotaryEmbedding: inv_freq = theta^(-2i/hd), angles in fp32)"""
        key = str(device)
        cur = self._rope.get(key)
        if cur is None or cur[0].shape[0] < n_pos:
            hd = self.config.head_dim
            n = max(n_pos, 1024)
            inv = 1.0 / (self.config.rope_theta ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
            fr = torch.arange(n, dtype=torch.float32)[:, None] * inv[None, :]
            self._rope[key] = (fr.cos().contiguous().to(device), fr.sin().contiguous().to(device))
        return self._rope[key]

    def embed(self, input_ids: torch.Tensor) -> torch.Tensor:
        """embed_tokens(ids) * sqrt(hidden) (pi0_arch.py:247-250; the 4.51 GemmaModel embedding is unscaled).
        Differentiable: the gather is the splice kernel with a tokens-only plan (its backward adds rows into the
        fp32 embedding gradient), the scale is applied in the tensor's dtype."""
        st = self.store
        d = self.config.hidden_size
        dummy = torch.zeros((1, d), device=st.device, dtype=st.compute_dtype)
        rows = Fn.SpliceFn.apply(dummy, st.params[self.embed_name], st, self.embed_name,
                                 input_ids.reshape(-1).to(device=st.device, dtype=torch.int64).contiguous())
        return Fn.ScaleFn.apply(rows, float(d) ** 0.5).view(*input_ids.shape, d)

    def pre_attention(self, x2d: torch.Tensor, li: int) -> torch.Tensor:
        """[M, d] -> fused qkv [M, (Hq + 2 Hkv) * hd] of layer li (input_layernorm + q/k/v projections)"""
        c, ly, st = self.config, self.layer_names[li], self.store
        nq = (c.num_attention_heads + 2 * c.num_key_value_heads) * c.head_dim
        h, _ = K.rmsnorm_fwd(x2d, self.norm_weight(ly["ln1"]), c.rms_norm_eps)
        return K.mm_nt(h, st.w(*ly["qkv"], shape=(nq, c.hidden_size)))

    def post_attention(self, x2d: torch.Tensor, attn2d: torch.Tensor, li: int) -> torch.Tensor:
        """x + o_proj(attn); then + down(gelu_tanh(gate) * up) of the post-attention norm"""
        c, ly, st = self.config, self.layer_names[li], self.store
        r = K.mm_nt(attn2d, st.w(ly["o"]), residual=x2d)
        h, _ = K.rmsnorm_fwd(r, self.norm_weight(ly["ln2"]), c.rms_norm_eps)
        gu = K.mm_nt(h, st.w(*ly["gu"], shape=(2 * c.intermediate_size, c.hidden_size)))
        return K.mm_nt(K.glu_fwd(gu, L.ACT_GELU_TANH), st.w(ly["down"]), residual=r)

    def final_norm(self, x2d: torch.Tensor) -> torch.Tensor:
        y, _ = K.rmsnorm_fwd(x2d, self.norm_weight(self.p + "norm.weight"), self.config.rms_norm_eps)
        return y
