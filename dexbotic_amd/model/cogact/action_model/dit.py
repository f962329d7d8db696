"""DiT action-noise predictor on libdexbotic_amd kernels (fp32).

Mirrors dexbotic/model/cogact/action_model/dit.py: TimestepEmbedder (:22-64), LabelEmbedder with CFG
token drop (:67-103), ActionEmbedder (:110-118), DiTBlock (:137-162; timm Attention(qkv_bias=True) /
Mlp(GELU tanh), LayerNorm without affine, eps 1e-6), FinalLayer (:165-178), DiT.forward (:273-292) and
forward_with_cfg (:294-311).  Parameter names = the reference's (``net.blocks.{k}.attn.qkv.weight`` ...).
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn as nn

from .... import _lib as L
from .... import functional as Fn
from ....engine import Fp32View, ParamStore


class DiT(nn.Module):
    def __init__(self, store: ParamStore, prefix: str, in_channels: int = 7, hidden_size: int = 1152,
                 depth: int = 28, num_heads: int = 16, mlp_ratio: float = 4.0, class_dropout_prob: float = 0.1,
                 token_size: int = 4096, future_action_window_size: int = 1, past_action_window_size: int = 0,
                 learn_sigma: bool = False):
        super().__init__()
        assert past_action_window_size == 0, "Error: action_history is not used now"
        assert not learn_sigma
        self.store, self.p = store, prefix
        self.in_channels = self.out_channels = in_channels
        self.hidden_size, self.depth, self.num_heads = hidden_size, depth, num_heads
        self.mlp_hidden = int(hidden_size * mlp_ratio)
        self.class_dropout_prob = class_dropout_prob
        self.token_size = token_size
        self.future_action_window_size = future_action_window_size
        self.T = future_action_window_size + 1          # chunk size; sequence = T + 1 with the condition token
        self.frequency_embedding_size = 256
        h, A, p = hidden_size, in_channels, prefix
        store.new_bucket()
        store.register([(p + "positional_embedding", (self.T + 1, h))])
        store.register([(p + "history_embedder.linear.weight", (h, A)), (p + "history_embedder.linear.bias", (h,))])
        store.register([(p + "x_embedder.linear.weight", (h, A)), (p + "x_embedder.linear.bias", (h,))])
        store.register([(p + "t_embedder.mlp.0.weight", (h, 256)), (p + "t_embedder.mlp.0.bias", (h,))])
        store.register([(p + "t_embedder.mlp.2.weight", (h, h)), (p + "t_embedder.mlp.2.bias", (h,))])
        store.register([(p + "z_embedder.uncondition", (1, token_size))])
        store.register([(p + "z_embedder.linear.weight", (h, token_size)), (p + "z_embedder.linear.bias", (h,))])
        self.block_specs = []
        for k in range(depth):
            b = f"{p}blocks.{k}."
            store.register([(b + "attn.qkv.weight", (3 * h, h))])
            store.register([(b + "attn.qkv.bias", (3 * h,))])
            store.register([(b + "attn.proj.weight", (h, h)), (b + "attn.proj.bias", (h,))])
            store.register([(b + "mlp.fc1.weight", (self.mlp_hidden, h)), (b + "mlp.fc1.bias", (self.mlp_hidden,))])
            store.register([(b + "mlp.fc2.weight", (h, self.mlp_hidden)), (b + "mlp.fc2.bias", (h,))])
            self.block_specs.append(Fn.VitBlockSpec(
                ln1_w=None, ln1_b=None, qkv_w=(b + "attn.qkv.weight",), qkv_b=(b + "attn.qkv.bias",),
                out_w=b + "attn.proj.weight", out_b=b + "attn.proj.bias", ln2_w=None, ln2_b=None,
                fc1_w=b + "mlp.fc1.weight", fc1_b=b + "mlp.fc1.bias", fc2_w=b + "mlp.fc2.weight",
                fc2_b=b + "mlp.fc2.bias", act=L.ACT_GELU_TANH, eps=1e-6, H=num_heads, D=h // num_heads,
                I=self.mlp_hidden))
        store.register([(p + "final_layer.linear.weight", (A, h)), (p + "final_layer.linear.bias", (A,))])
        self._freqs = {}
        self.training_mode_drop = True

    # ---- init (dit.py:244-271) -------------------------------------------------------------------------
    @torch.no_grad()
    def initialize_weights(self) -> None:
        st, p = self.store, self.p
        for name, s in st.slots.items():
            if not name.startswith(p):
                continue
            t = st.w32(name)
            if name.endswith(".bias"):
                t.zero_()
            elif name.endswith("linear.weight") or ".attn." in name or ".mlp.fc" in name:
                if t.dim() == 2:
                    nn.init.xavier_uniform_(t)
        h = self.hidden_size
        st.w32(p + "positional_embedding").copy_(h ** -0.5 * torch.randn(self.T + 1, h))
        for n in ("x_embedder.linear.weight", "history_embedder.linear.weight", "z_embedder.uncondition",
                  "z_embedder.linear.weight", "t_embedder.mlp.0.weight", "t_embedder.mlp.2.weight"):
            nn.init.normal_(st.w32(p + n), std=0.02)
        st.w32(p + "final_layer.linear.weight").zero_()
        st.w32(p + "final_layer.linear.bias").zero_()

    # ---- pieces ----------------------------------------------------------------------------------------
    def _anchor(self):
        return self.store.params[self.p + "final_layer.linear.weight"]

    def _timestep_freqs(self, device) -> torch.Tensor:
        key = str(device)
        if key not in self._freqs:
            half = self.frequency_embedding_size // 2
            # built with the same torch ops as the reference (dit.py:45-49) so the table is bit-identical
            f = torch.exp(-math.log(10000) * torch.arange(start=0, end=half, dtype=torch.float32) / half)
            self._freqs[key] = f.to(device)
        return self._freqs[key]

    def forward(self, x: torch.Tensor, t: torch.Tensor, z: torch.Tensor, drop_ids: Optional[torch.Tensor] = None,
                train: Optional[bool] = None):
        """x (N,T,A) noisy actions, t (N,) timesteps, z (N,1,token) conditions -> eps_hat (N,T,A).
        ``drop_ids`` (N,) bool/uint8: classifier-free-guidance token drop (drawn by the caller in train mode)."""
        from .... import kernels as K
        st = Fp32View(self.store)
        p, h = self.p, self.hidden_size
        N, T, A = x.shape
        anchor = self._anchor()
        x = x.float().contiguous()
        xe = Fn.LinearFn.apply(x, anchor, st, p + "x_embedder.linear.weight", p + "x_embedder.linear.bias",
                               L.ACT_NONE, None)                                        # (N,T,h)
        tf = K.timestep_embedding(t.float().contiguous(), self._timestep_freqs(x.device))   # (N,256)
        te = Fn.MlpFn.apply(tf, anchor, st, p + "t_embedder.mlp.0.weight", p + "t_embedder.mlp.0.bias",
                            p + "t_embedder.mlp.2.weight", p + "t_embedder.mlp.2.bias", L.ACT_SILU)  # (N,h)
        z2 = z.reshape(N, self.token_size).float()
        if drop_ids is not None:
            z2 = Fn.TokenDropFn.apply(z2, anchor, st, p + "z_embedder.uncondition",
                                      drop_ids.to(torch.uint8).contiguous())
        ze = Fn.LinearFn.apply(z2, anchor, st, p + "z_embedder.linear.weight", p + "z_embedder.linear.bias",
                               L.ACT_NONE, None)                                        # (N,h)
        hcur = Fn.DitAssembleFn.apply(xe, te, ze, anchor, st, p + "positional_embedding")   # (N,T+1,h)
        for sp in self.block_specs:
            sp.N, sp.T = N, T + 1
            hcur = Fn.VitBlockFn.apply(hcur, anchor, st, sp)
        hcur = Fn.NormFn.apply(hcur.reshape(N * (T + 1), h), anchor, st, "ln", None, None, 1e-6)
        out = Fn.LinearFn.apply(hcur, anchor, st, p + "final_layer.linear.weight", p + "final_layer.linear.bias",
                                L.ACT_NONE, None)
        return out.view(N, T + 1, A)[:, 1:, :]

    def forward_with_cfg(self, x, t, z, cfg_scale=None):
        """dit.py:294-311: both halves of the batch are the FIRST half of x; returns the RAW network output
        for [cond; uncond] — the guidance mix eps = u + s (c - u) is fused into dxa_ddim_step
        (diffusion.SpacedDiffusion.ddim_sample_loop receives cfg_scale)."""
        half = x[: len(x) // 2]
        return self.forward(torch.cat([half, half], dim=0), t, z)
