"""Qwen2 decoder backbone on libdexbotic_amd kernels.

Stands in for the HF ``Qwen2Model`` the reference instantiates through ``AutoModel.from_config``
(dexbotic/model/dexbotic_arch.py:52-62) and calls at cogact_arch.py:97-106; arithmetic per
HF:qwen2/modeling_qwen2.py (RMSNorm :238-253, attention + rotate-half RoPE :105-235, SwiGLU :35-48,
final norm — ``hidden_states[-1]`` is post-norm).  Parameter names are HF's.
"""
from __future__ import annotations

from dataclasses import dataclass, asdict
from typing import Sequence, Optional

import torch
import torch.nn as nn

from ... import functional as Fn
from ... import kernels as K
from ...engine import ParamStore


@dataclass
class Qwen2Config:
    """subset of HF Qwen2Config that defines the arithmetic (defaults = Qwen2.5-7B)"""
    vocab_size: int = 152064
    hidden_size: int = 3584
    intermediate_size: int = 18944
    num_hidden_layers: int = 28
    num_attention_heads: int = 28
    num_key_value_heads: int = 4
    rms_norm_eps: float = 1e-6
    rope_theta: float = 1e6
    max_position_embeddings: int = 32768
    model_type: str = "qwen2"

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    def to_dict(self):
        return asdict(self)

    @classmethod
    def from_any(cls, obj) -> "Qwen2Config":
        if isinstance(obj, cls):
            return obj
        d = obj if isinstance(obj, dict) else (obj.to_dict() if hasattr(obj, "to_dict") else vars(obj))
        rp = d.get("rope_parameters") or {}
        theta = d.get("rope_theta", rp.get("rope_theta", 1e6))
        keys = {f for f in cls.__dataclass_fields__}
        kw = {k: v for k, v in d.items() if k in keys and v is not None}
        kw["rope_theta"] = theta
        if kw.get("model_type", "qwen2") != "qwen2":
            raise NotImplementedError(f"llm_config.model_type={kw['model_type']!r}: only the Qwen2 backbone of "
                                      "DB-CogACT is implemented natively")
        return cls(**kw)


class Qwen2Backbone(nn.Module):
    def __init__(self, store: ParamStore, prefix: str, config: Qwen2Config):
        super().__init__()
        self.store, self.p, self.config = store, prefix, config
        c = config
        d, f, hd = c.hidden_size, c.intermediate_size, c.head_dim
        Hq, Hkv = c.num_attention_heads, c.num_key_value_heads
        store.new_bucket()
        store.register([(prefix + "embed_tokens.weight", (c.vocab_size, d))])
        self.layer_specs = []
        for i in range(c.num_hidden_layers):
            lp = f"{prefix}layers.{i}."
            store.new_bucket()
            qkv_w = tuple(lp + f"self_attn.{n}_proj.weight" for n in "qkv")
            qkv_b = tuple(lp + f"self_attn.{n}_proj.bias" for n in "qkv")
            store.register([(lp + "input_layernorm.weight", (d,))])
            store.register([(qkv_w[0], (Hq * hd, d)), (qkv_w[1], (Hkv * hd, d)), (qkv_w[2], (Hkv * hd, d))])
            store.register([(qkv_b[0], (Hq * hd,)), (qkv_b[1], (Hkv * hd,)), (qkv_b[2], (Hkv * hd,))])
            store.register([(lp + "self_attn.o_proj.weight", (d, Hq * hd))])
            store.register([(lp + "post_attention_layernorm.weight", (d,))])
            gu = (lp + "mlp.gate_proj.weight", lp + "mlp.up_proj.weight")
            store.register([(gu[0], (f, d)), (gu[1], (f, d))])
            store.register([(lp + "mlp.down_proj.weight", (d, f))])
            self.layer_specs.append(Fn.Qwen2LayerSpec(
                ln1=lp + "input_layernorm.weight", qkv_w=qkv_w, qkv_b=qkv_b, o_w=lp + "self_attn.o_proj.weight",
                ln2=lp + "post_attention_layernorm.weight", gu_w=gu, down_w=lp + "mlp.down_proj.weight",
                Hq=Hq, Hkv=Hkv, D=hd, d=d, F=f, eps=c.rms_norm_eps))
        store.new_bucket()
        store.register([(prefix + "norm.weight", (d,))])
        self._rope = {}

    @property
    def embed_name(self) -> str:
        return self.p + "embed_tokens.weight"

    @property
    def vocab_size(self) -> int:
        return self.config.vocab_size

    def rope_tables(self, S: int, device):
        """cos/sin [S, hd/2] fp32 built with the same torch ops as Qwen2RotaryEmbedding
        (HF:qwen2/modeling_qwen2.py:52-104; position_ids = arange(S), dexbotic_arch.py:251)."""
        key = (S, str(device))
        if key not in self._rope:
            hd = self.config.head_dim
            inv_freq = 1.0 / (self.config.rope_theta ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
            freqs = torch.arange(S, dtype=torch.float32)[:, None] * inv_freq[None, :]
            self._rope[key] = (freqs.cos().contiguous().to(device), freqs.sin().contiguous().to(device))
        return self._rope[key]

    def forward(self, inputs_embeds: torch.Tensor, kv_start: Optional[torch.Tensor] = None,
                kv_end: Optional[torch.Tensor] = None) -> torch.Tensor:
        """inputs_embeds [B,S,d] (compute dtype) -> last hidden state [B,S,d] (post final RMSNorm)."""
        B, S, d = inputs_embeds.shape
        cos_t, sin_t = self.rope_tables(S, inputs_embeds.device)
        st = self.store
        x = inputs_embeds.reshape(B * S, d)
        grad_mode = torch.is_grad_enabled()      # (read HERE: inside an autograd Function's forward it is always off)
        for sp in self.layer_specs:
            sp.B, sp.S, sp.grad_mode = B, S, grad_mode
            x = Fn.Qwen2LayerFn.apply(x, st.params[sp.ln1], st, sp, cos_t, sin_t, kv_start, kv_end)
        x = Fn.NormFn.apply(x, st.params[self.p + "norm.weight"], st, "rms", self.p + "norm.weight", None,
                            self.config.rms_norm_eps)
        return x.view(B, S, d)


    # ------------------------------------------------------------------------------- KV-cached inference
    def new_cache(self, batch: int, max_len: int, device, dtype) -> "KVCache":
        c = self.config
        return KVCache(c.num_hidden_layers, batch, c.num_key_value_heads, max_len, c.head_dim, device, dtype)

    @torch.no_grad()
    def forward_cached(self, inputs_embeds: torch.Tensor, cache: "KVCache", pad: Optional[Sequence[int]] = None) -> torch.Tensor:
        """Prefill (S > 1) or decode (S = 1) step over a key/value cache: the use_cache=True path of HF Qwen2Model
        that GenerationMixin.generate drives (discrete_vla_arch.py:33-41).  Post-RoPE keys and values of every
        layer are appended at positions [cache.length, cache.length + S); attention is causal over the whole cache
        (queries sit at the END of the key range).
        ``pad``: per-sample count of LEFT padding slots in front of the prompt (prompts of unequal length in one batch,
        right-aligned like HF's generate wants them): sample b attends to the keys [pad[b], total) only and its rotary
        positions count from its first real token, position = slot - pad[b] — what HF derives from the attention mask
        (GenerationMixin: position_ids = attention_mask.cumsum(-1) - 1)."""
        B, S, d = inputs_embeds.shape
        past, total = cache.length, cache.length + S
        if total > cache.max_len:
            raise ValueError(f"KV cache of {cache.max_len} positions cannot take {total}")
        if B == 1 and S == 1 and past > 0 and inputs_embeds.dtype == torch.bfloat16 and inputs_embeds.is_cuda:
            fused = self._decode_state(cache)
            if fused is not None:
                # ONE persistent launch for the whole decoder pass of the new token (csrc/decode_fused.hip)
                kv_lo = int(pad[0]) if pad is not None else 0
                cos_m, sin_m = self.rope_tables(cache.max_len, inputs_embeds.device)
                c = self.config
                out = torch.empty(d, device=inputs_embeds.device, dtype=torch.bfloat16)
                K.decode_step(fused["table"], inputs_embeds.reshape(d).contiguous(), out, fused["final_w"], cos_m[past - kv_lo],
                              sin_m[past - kv_lo], fused["ws"], c.num_hidden_layers, d, c.num_attention_heads,
                              c.num_key_value_heads, c.head_dim, c.intermediate_size, past, kv_lo, cache.max_len, c.rms_norm_eps)
                cache.length = total
                cache.fused_steps += 1
                return out.view(1, 1, d)
        cos_all, sin_all = self.rope_tables(total, inputs_embeds.device)
        cos_t, sin_t = cos_all[past:total], sin_all[past:total]
        pos = kv_start = kv_end = None
        if pad is not None and any(int(p_) != 0 for p_ in pad):
            dev = inputs_embeds.device
            padt = torch.tensor([int(p_) for p_ in pad], dtype=torch.int32, device=dev)
            slots = torch.arange(past, total, dtype=torch.int32, device=dev)
            pos = (slots.view(1, S) - padt.view(B, 1)).clamp_(min=0).reshape(-1).contiguous()      # row of the full tables
            cos_t, sin_t = cos_all, sin_all
            kv_start = padt
            kv_end = torch.full((B,), total, dtype=torch.int32, device=dev)
        st = self.store
        x = inputs_embeds.reshape(B * S, d).contiguous()
        for i, sp in enumerate(self.layer_specs):
            Hq, Hkv, D, F_ = sp.Hq, sp.Hkv, sp.D, sp.F
            nq = (Hq + 2 * Hkv) * D
            h1, _ = K.rmsnorm_fwd(x, st.w(sp.ln1), sp.eps)
            qkv = K.mm_nt(h1, st.w(*sp.qkv_w, shape=(nq, d)), bias=st.w(*sp.qkv_b, shape=(nq,)))
            q, k, v = K.rope_split(qkv, cos_t, sin_t, pos, B, S, Hq, Hkv, D)
            cache.k[i][:, :, past:total].copy_(k)
            cache.v[i][:, :, past:total].copy_(v)
            o = torch.empty((B, S, Hq, D), device=x.device, dtype=x.dtype)
            K.attn_fwd(q, cache.k[i][:, :, :total], cache.v[i][:, :, :total], o.permute(0, 2, 1, 3), causal=True,
                       scale=D ** -0.5, kv_start=kv_start, kv_end=kv_end)
            x2 = K.mm_nt(o.view(B * S, Hq * D), st.w(sp.o_w), residual=x)
            h2, _ = K.rmsnorm_fwd(x2, st.w(sp.ln2), sp.eps)
            w_gu = st.w(*sp.gu_w, shape=(2 * F_, d))
            if K.swiglu_gemm_supported(h2, w_gu):
                a, _ = K.mm_nt_swiglu(h2, w_gu, keep_pre=False)
            else:
                a = K.swiglu_fwd(K.mm_nt(h2, w_gu))
            x = K.mm_nt(a, st.w(sp.down_w), residual=x2)
        x, _ = K.rmsnorm_fwd(x, st.w(self.p + "norm.weight"), self.config.rms_norm_eps)
        cache.length = total
        return x.view(B, S, d)


    def _decode_state(self, cache: "KVCache"):
        """pointer table, final-norm weight and workspace of the persistent decode step for this cache (built at the first
        single-token pass, kept on the cache), or None where the launch does not apply (DXA_DECODE_FUSED=0, widths outside the
        kernel's limits, a weight that is not 16-byte aligned, fp32 weights): the per-op path above then runs."""
        import os
        st, c = self.store, self.config
        key = st.weights_key()
        if cache.fused is not None and cache.fused.get("key") == key:
            return cache.fused["state"]
        state = None
        d, F_, D = c.hidden_size, c.intermediate_size, c.head_dim
        Hq, Hkv = c.num_attention_heads, c.num_key_value_heads
        if (os.environ.get("DXA_DECODE_FUSED", "1") != "0" and st.shadow is not None and cache.batch == 1 and
                K.decode_step_supported(d, Hq, Hkv, D, F_) and cache.max_len - 1 <= K.DECODE_FUSED_MAX_KEYS):
            nq = (Hq + 2 * Hkv) * D
            ptrs = []
            for i, sp in enumerate(self.layer_specs):
                ptrs += [st.w(sp.ln1).data_ptr(), st.w(*sp.qkv_w, shape=(nq, d)).data_ptr(), st.w(*sp.qkv_b, shape=(nq,)).data_ptr(),
                         st.w(sp.o_w).data_ptr(), st.w(sp.ln2).data_ptr(), st.w(*sp.gu_w, shape=(2 * F_, d)).data_ptr(),
                         st.w(sp.down_w).data_ptr(), cache.k[i].data_ptr(), cache.v[i].data_ptr()]
            final_w = st.w(self.p + "norm.weight")
            if all(p_ % 16 == 0 for p_ in ptrs) and final_w.data_ptr() % 16 == 0:
                dev = cache.k[0].device
                state = {"table": torch.tensor(ptrs, dtype=torch.int64).to(dev), "final_w": final_w,
                         "ws": K.decode_step_workspace(d, Hq, Hkv, D, F_, dev)}
        cache.fused = {"key": key, "state": state}
        return state


class KVCache:
    """per-layer post-RoPE keys / values, head-major [B, Hkv, max_len, D] (the layout the attention kernels read)"""

    def __init__(self, layers: int, batch: int, kv_heads: int, max_len: int, head_dim: int, device, dtype):
        self.k = [torch.empty((batch, kv_heads, max_len, head_dim), device=device, dtype=dtype) for _ in range(layers)]
        self.v = [torch.empty((batch, kv_heads, max_len, head_dim), device=device, dtype=dtype) for _ in range(layers)]
        self.max_len = max_len
        self.batch = batch
        self.length = 0
        self.fused = None          # Qwen2Backbone._decode_state: pointer table / workspace of the persistent decode step
        self.fused_steps = 0       # single-token passes that ran as one persistent launch
