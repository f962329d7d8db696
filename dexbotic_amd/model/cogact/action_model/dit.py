"""DiT action-noise predictor on libdexbotic_amd kernels (fp32).

Mirrors dexbotic/model/cogact/action_model/dit.py: TimestepEmbedder (:22-64), LabelEmbedder with CFG
token drop (:67-103), ActionEmbedder (:110-118), DiTBlock (:137-162; timm Attention(qkv_bias=True) /
Mlp(GELU tanh), LayerNorm without affine, eps 1e-6), FinalLayer (:165-178), DiT.forward (:273-292) and
forward_with_cfg (:294-311).  Parameter names = the reference's (``net.blocks.{k}.attn.qkv.weight`` ...).
"""
from __future__ import annotations

import math
import os
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
                 learn_sigma: bool = False, use_per_attn: bool = False, per_token_size: Optional[int] = None):
        """``use_per_attn``: the MemVLA variant (memvla/action_model/dit.py:136-185, 240-249) — every block also
        cross-attends to the perceptual tokens (nn.MultiheadAttention + affine norm3), a PerTokenEmbedder replaces
        the unused HistoryEmbedder."""
        super().__init__()
        self.use_per_attn, self.per_token_size = use_per_attn, per_token_size
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
        if use_per_attn:
            assert per_token_size is not None
            store.register([(p + "per_token_embedder.linear.weight", (h, per_token_size)),
                            (p + "per_token_embedder.linear.bias", (h,))])
        else:
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
            if use_per_attn:
                store.register([(b + "per_attn.in_proj_weight", (3 * h, h)), (b + "per_attn.in_proj_bias", (3 * h,))])
                store.register([(b + "per_attn.out_proj.weight", (h, h)), (b + "per_attn.out_proj.bias", (h,))])
                store.register([(b + "norm3.weight", (h,)), (b + "norm3.bias", (h,))], layernorm=True)
            self.block_specs.append(Fn.VitBlockSpec(
                ln1_w=None, ln1_b=None, qkv_w=(b + "attn.qkv.weight",), qkv_b=(b + "attn.qkv.bias",),
                out_w=b + "attn.proj.weight", out_b=b + "attn.proj.bias", ln2_w=None, ln2_b=None,
                fc1_w=b + "mlp.fc1.weight", fc1_b=b + "mlp.fc1.bias", fc2_w=b + "mlp.fc2.weight",
                fc2_b=b + "mlp.fc2.bias", act=L.ACT_GELU_TANH, eps=1e-6, H=num_heads, D=h // num_heads,
                I=self.mlp_hidden, ckpt=False))
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
        emb = "per_token_embedder.linear.weight" if self.use_per_attn else "history_embedder.linear.weight"
        for n in ("x_embedder.linear.weight", emb, "z_embedder.uncondition",
                  "z_embedder.linear.weight", "t_embedder.mlp.0.weight", "t_embedder.mlp.2.weight"):
            nn.init.normal_(st.w32(p + n), std=0.02)
        if self.use_per_attn:                       # zero-initialised cross attention (memvla dit.py:167-171)
            for k in range(self.depth):
                b = f"{p}blocks.{k}."
                for n in ("per_attn.in_proj_weight", "per_attn.in_proj_bias", "per_attn.out_proj.weight",
                          "per_attn.out_proj.bias", "norm3.bias"):
                    st.w32(b + n).zero_()
                st.w32(b + "norm3.weight").fill_(1.0)
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

    def _block_with_per_attn(self, st, k: int, hcur: torch.Tensor, pe: torch.Tensor, N: int, T1: int,
                             kv: Optional[torch.Tensor] = None, R: int = 1) -> torch.Tensor:
        """memvla DiTBlock (memvla/action_model/dit.py:175-187): x + attn(norm1 x); x + MHA(norm3 x, per, per);
        x + mlp(norm2 x) — composed from the small autograd pieces (the fused VitBlockFn has no slot for the middle
        term).  nn.MultiheadAttention's packed in_proj: rows [:h] on the queries, rows [h:] on the perceptual tokens
        (functional.PackedInProjFn).  ``kv``: the sampler's per-request [k | v] of this block (precompute_per_kv).
        ``R`` > 1 (forward(per_repeat=R)): ``pe`` holds the N / R distinct samples and the rows of ``hcur`` are sample-major
        with a sample's R diffusion repeats adjacent — the perceptual attention is not causal, so the R x T1 queries of a
        sample form one query sequence over that sample's keys."""
        b, h, H = f"{self.p}blocks.{k}.", self.hidden_size, self.num_heads
        D, anchor = h // H, self._anchor()
        if kv is not None:
            if torch.is_grad_enabled():
                raise RuntimeError("DiT: per_kv (precompute_per_kv) is an inference-time cache: its products are not recorded "
                                   "for autograd; call forward(per_token=...) when gradients are needed")
            return self._block_with_per_attn_sampler(st, b, hcur, kv, N, T1)
        lin = lambda x, wn, bn: Fn.LinearFn.apply(x, anchor, st, wn, bn, L.ACT_NONE, None)
        # every tensor with two consumers is forked explicitly and the packed projections stay packed through the attention
        # (Fn.ForkFn / Fn.AttnPackedFn): the backward then consists of library launches only
        hn, hr = Fn.ForkFn.apply(hcur)
        y = Fn.NormFn.apply(hn, anchor, st, "ln", None, None, 1e-6)
        qkv = lin(y, b + "attn.qkv.weight", b + "attn.qkv.bias").view(N, T1, 3, H, D)
        o = Fn.AttnPackedFn.apply(None, qkv).reshape(N * T1, h)
        hcur = Fn.AddFn.apply(hr, lin(o, b + "attn.proj.weight", b + "attn.proj.bias"))
        hn, hr = Fn.ForkFn.apply(hcur)
        y3 = Fn.NormFn.apply(hn, anchor, st, "ln", b + "norm3.weight", b + "norm3.bias", 1e-6)
        P_ = pe.shape[1]
        Nk = N // R
        qf, kvf = Fn.PackedInProjFn.apply(y3, pe.reshape(Nk * P_, h), anchor, st, b + "per_attn.in_proj_weight",
                                          b + "per_attn.in_proj_bias", h)            # q [N*T1, h], [k | v] [Nk*P, 2h]
        o2 = Fn.AttnPackedFn.apply(qf.view(Nk, R * T1, H, D), kvf.view(Nk, P_, 2, H, D)).reshape(N * T1, h)
        hcur = Fn.AddFn.apply(hr, lin(o2, b + "per_attn.out_proj.weight", b + "per_attn.out_proj.bias"))
        hn, hr = Fn.ForkFn.apply(hcur)
        y2 = Fn.NormFn.apply(hn, anchor, st, "ln", None, None, 1e-6)
        m = Fn.MlpFn.apply(y2, anchor, st, b + "mlp.fc1.weight", b + "mlp.fc1.bias", b + "mlp.fc2.weight",
                           b + "mlp.fc2.bias", L.ACT_GELU_TANH)
        return Fn.AddFn.apply(hr, m)

    def _block_with_per_attn_sampler(self, st, b: str, hcur: torch.Tensor, kv: torch.Tensor, N: int, T1: int) -> torch.Tensor:
        """the same block for the sampler (no autograd graph, cached perceptual [k | v]): 11 launches instead of 14 — the three
        residual additions ride in the epilogues of the projections that produce the addend (same two fp32 additions in the
        same order as the stand-alone add).  A launch costs >= 4.5 us on this part however small the kernel; the sampler is
        240 such blocks a frame."""
        from .... import kernels as K
        h, H = self.hidden_size, self.num_heads
        D = h // H
        W = lambda n: st.w(b + n)

        def attend(q, k, v):
            o = torch.empty((N, T1, H, D), device=hcur.device, dtype=hcur.dtype)
            P = lambda t: t.permute(0, 2, 1, 3)
            K.attn_fwd(P(q), P(k), P(v), P(o), causal=False, scale=D ** -0.5)
            return o.view(N * T1, h)
        y = K.layernorm_fwd(hcur, None, None, 1e-6)[0]
        qkv = K.mm_nt(y, W("attn.qkv.weight"), bias=W("attn.qkv.bias")).view(N, T1, 3, H, D)
        hcur = K.mm_nt(attend(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]), W("attn.proj.weight"), bias=W("attn.proj.bias"),
                       residual=hcur)
        y3 = K.layernorm_fwd(hcur, W("norm3.weight"), W("norm3.bias"), 1e-6)[0]
        q = K.mm_nt(y3, W("per_attn.in_proj_weight")[:h], bias=W("per_attn.in_proj_bias")[:h]).view(N, T1, H, D)
        hcur = K.mm_nt(attend(q, kv[:, :, 0], kv[:, :, 1]), W("per_attn.out_proj.weight"), bias=W("per_attn.out_proj.bias"),
                       residual=hcur)
        y2 = K.layernorm_fwd(hcur, None, None, 1e-6)[0]
        a = K.mm_nt(y2, W("mlp.fc1.weight"), bias=W("mlp.fc1.bias"), act=L.ACT_GELU_TANH)
        return K.mm_nt(a, W("mlp.fc2.weight"), bias=W("mlp.fc2.bias"), residual=hcur)

    @torch.no_grad()
    def precompute_per_kv(self, per_token: torch.Tensor, packed: bool = False):
        """inference: the perceptual-token embedding and every block's key/value projection of it depend on the request, not on
        the DDIM step — the sampler computes them once ([N, P, 2, H, D] per block; DiT-L: 24 x [N*P, 1024] x [2048, 1024]^T
        products a step otherwise) and hands them to forward(per_kv=...).  Same kernels on the same operands as the per-step
        path: the samples are bit-identical.  ``packed``: ONE tensor [depth, N, P, 2, h] (what the one-launch sampler takes)."""
        from .... import kernels as K
        st = Fp32View(self.store)
        p, h, H = self.p, self.hidden_size, self.num_heads
        N, P_ = per_token.shape[:2]
        pe = K.mm_nt(per_token.float().reshape(N * P_, -1).contiguous(), st.w(p + "per_token_embedder.linear.weight"),
                     bias=st.w(p + "per_token_embedder.linear.bias"))
        if packed:
            kv = torch.empty((self.depth, N * P_, 2 * h), device=pe.device, dtype=torch.float32)
            for k in range(self.depth):
                K.mm_nt(pe, st.w(f"{p}blocks.{k}.per_attn.in_proj_weight")[h:], bias=st.w(f"{p}blocks.{k}.per_attn.in_proj_bias")[h:],
                        out=kv[k])
            return kv.view(self.depth, N, P_, 2, h)
        return [K.mm_nt(pe, st.w(f"{p}blocks.{k}.per_attn.in_proj_weight")[h:],
                        bias=st.w(f"{p}blocks.{k}.per_attn.in_proj_bias")[h:]).view(N, P_, 2, H, h // H)
                for k in range(self.depth)]

    def _use_fused_blocks(self, N: int, T1: int) -> bool:
        from .... import kernels as K
        # one persistent launch for all blocks of a single-request denoising call (csrc/dit_fused.hip);
        # DXA_DIT_FUSED=0 restores the block-by-block kernels
        return (not torch.is_grad_enabled() and os.environ.get("DXA_DIT_FUSED", "1") != "0" and
                getattr(self, "allow_fused", True) and self.store.device.type == "cuda" and
                K.dit_blocks_supported(N, T1, self.hidden_size, self.num_heads, self.mlp_hidden))

    def _weight_table(self, st) -> torch.Tensor:
        """device array of the raw fp32 master pointers of every block (the arena never moves); with perceptual attention 14 per
        block (dxa_dit_bf16_pack_per's order)"""
        if getattr(self, "_wtab", None) is None:
            ptrs = []
            names = ("attn.qkv.weight", "attn.qkv.bias", "attn.proj.weight", "attn.proj.bias", "mlp.fc1.weight",
                     "mlp.fc1.bias", "mlp.fc2.weight", "mlp.fc2.bias")
            if self.use_per_attn:
                names += ("per_attn.in_proj_weight", "per_attn.in_proj_bias", "per_attn.out_proj.weight", "per_attn.out_proj.bias",
                          "norm3.weight", "norm3.bias")
            for k in range(self.depth):
                b = f"{self.p}blocks.{k}."
                for n in names:
                    w = st.w(b + n)
                    assert w.dtype == torch.float32 and w.is_contiguous()
                    ptrs.append(w.data_ptr())
            self._wtab = torch.tensor(ptrs, dtype=torch.int64).to(self.store.device)
        return self._wtab

    def _bf16_sampler(self, N: int, T1: int) -> bool:
        """the one-launch sampler multiplies in bf16 when the model is served in bfloat16 (DXA_DIT_BF16=0: exact fp32 products)"""
        from .... import kernels as K
        return (self.store.compute_dtype == torch.bfloat16 and os.environ.get("DXA_DIT_BF16", "1") != "0" and
                K.dit_sample_bf16_supported(N, T1, self.hidden_size, self.num_heads, self.mlp_hidden))

    def _packed_table(self, st) -> torch.Tensor:
        """bf16 operand copy of the blocks' matrices (K.dit_bf16_pack), re-packed INTO THE SAME arena whenever the fp32 masters
        may have moved (ParamStore.weights_key): a captured graph that holds its pointers stays valid.  The first build allocates:
        not under stream capture (the first eager request of a shape builds it)."""
        from .... import kernels as K
        key = self.store.weights_key()
        ent = getattr(self, "_bf16_pack", None)
        if ent is None:
            arena, table = K.dit_bf16_pack(self._weight_table(st), self.depth, self.hidden_size, self.mlp_hidden, per=self.use_per_attn)
            self._bf16_pack = ent = (key, arena, table)
        elif ent[0] != key:
            K.dit_bf16_pack(self._weight_table(st), self.depth, self.hidden_size, self.mlp_hidden, out=(ent[1], ent[2]),
                            per=self.use_per_attn)
            self._bf16_pack = ent = (key, ent[1], ent[2])
        return ent[2]

    def refresh_packed(self) -> None:
        """before a graph replay (no Python of this module runs in it): bring an existing packed copy up to date"""
        if getattr(self, "_bf16_pack", None) is not None:
            self._packed_table(Fp32View(self.store))

    def forward(self, x: torch.Tensor, t: torch.Tensor, z: torch.Tensor, drop_ids: Optional[torch.Tensor] = None,
                train: Optional[bool] = None, per_token: Optional[torch.Tensor] = None, per_kv: Optional[list] = None,
                per_repeat: int = 1):
        """x (N,T,A) noisy actions, t (N,) timesteps, z (N,1,token) conditions -> eps_hat (N,T,A).
        ``drop_ids`` (N,) bool/uint8: classifier-free-guidance token drop (drawn by the caller in train mode).
        ``per_repeat`` = R > 1: the batch is R diffusion repeats of N / R samples stacked repeat-major (x.repeat(R, 1, 1) order,
        memvla_arch.py:515-519) and ``per_token`` holds the N / R DISTINCT perceptual sequences — the reference repeats them R
        times and projects every copy in each of the blocks (24 x [R N/R P, 1024] x [2048, 1024]^T on identical rows); here the
        embedding and the per-block key/value projections run once per distinct sample.  The samples are walked sample-major
        inside (one row permutation of the tiny inputs, undone on the output): same values, R x fewer projection rows."""
        from .... import kernels as K
        st = Fp32View(self.store)
        p, h = self.p, self.hidden_size
        N, T, A = x.shape
        anchor = self._anchor()
        R = int(per_repeat) if (self.use_per_attn and per_token is not None) else 1
        if R > 1:
            assert N % R == 0 and per_token.shape[0] == N // R, (N, R, tuple(per_token.shape))
            perm, inv = self._repeat_perm(N, R, x.device)
            x, t, z = x[perm], t[perm], z[perm]
            drop_ids = None if drop_ids is None else drop_ids[perm]
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
        if self.use_per_attn:
            assert per_token is not None or per_kv is not None
            pe = None if per_kv is not None else Fn.LinearFn.apply(
                per_token.float().contiguous(), anchor, st, p + "per_token_embedder.linear.weight",
                p + "per_token_embedder.linear.bias", L.ACT_NONE, None)                         # (N,P,h)
            hcur = hcur.reshape(N * (T + 1), h)
            pes = [None] * self.depth if pe is None else \
                (Fn.ForkFn.apply(pe, self.depth) if (torch.is_grad_enabled() and pe.requires_grad) else [pe] * self.depth)
            for k in range(self.depth):
                hcur = self._block_with_per_attn(st, k, hcur, pes[k], N, T + 1, None if per_kv is None else per_kv[k], R)
        elif self._use_fused_blocks(N, T + 1):
            # inference, one request: every block in ONE persistent launch (csrc/dit_fused.hip)
            self.used_fused = True
            self.store.wait_pending()                  # the kernel reads the masters through raw pointers (_weight_table)
            hcur = K.dit_blocks_fwd(hcur.reshape(N * (T + 1), h).contiguous(), self._weight_table(st), self.depth, N, T + 1, h,
                                    self.num_heads, self.mlp_hidden, 1e-6)
        else:
            for sp in self.block_specs:
                sp.N, sp.T = N, T + 1
                hcur = Fn.VitBlockFn.apply(hcur, anchor, st, sp)
        hcur = Fn.NormFn.apply(hcur.reshape(N * (T + 1), h), anchor, st, "ln", None, None, 1e-6)
        out = Fn.LinearFn.apply(hcur, anchor, st, p + "final_layer.linear.weight", p + "final_layer.linear.bias",
                                L.ACT_NONE, None)
        out = out.view(N, T + 1, A)[:, 1:, :]
        return out[inv] if R > 1 else out

    def _repeat_perm(self, N: int, R: int, device):
        """row n' = b R + r of the sample-major walk <- row n = r (N/R) + b of the caller's repeat-major batch, and its inverse"""
        key = (N, R, str(device))
        tabs = self.__dict__.setdefault("_perm_tabs", {})
        if key not in tabs:
            perm = torch.arange(N).view(R, N // R).t().reshape(-1)
            tabs[key] = (perm.to(device), torch.argsort(perm).to(device))
        return tabs[key]

    # ---- the whole sampler in one launch ---------------------------------------------------------------------------
    def fused_sampler_ok(self, N: int, T1: int, P: int = 0) -> bool:
        """``P``: perceptual keys per sample (MemVLA's DiT: only the bf16 sampler has the perceptual-attention phases)"""
        from .... import kernels as K
        if self.in_channels > 8 or os.environ.get("DXA_DIT_SAMPLER", "1") == "0":
            return False
        if self.use_per_attn:
            return (P > 0 and not torch.is_grad_enabled() and os.environ.get("DXA_DIT_FUSED", "1") != "0" and
                    getattr(self, "allow_fused", True) and self.store.device.type == "cuda" and
                    self.store.compute_dtype == torch.bfloat16 and os.environ.get("DXA_DIT_BF16", "1") != "0" and
                    K.dit_sample_bf16_supported(N, T1, self.hidden_size, self.num_heads, self.mlp_hidden, P))
        return self._use_fused_blocks(N, T1)

    def _sampler_tables(self, diffusion, device):
        """per (schedule, device): the timesteps in execution order and the three DDIM coefficients per step, rounded to fp32
        exactly like the per-step path (diffusion.ddim_sample_loop: float(np.float32(a[i])))"""
        key = (id(diffusion), str(device))
        tabs = self.__dict__.setdefault("_samp_tabs", {})
        if key not in tabs:
            import numpy as np
            order = list(reversed(range(diffusion.num_timesteps)))
            tv = torch.tensor([float(diffusion.timestep_map[i]) for i in order], dtype=torch.float32)
            coef = np.zeros((len(order), 4), dtype=np.float32)
            for r, i in enumerate(order):
                coef[r, 0] = np.float32(diffusion.sqrt_recip_alphas_cumprod[i])
                coef[r, 1] = np.float32(diffusion.sqrt_recipm1_alphas_cumprod[i])
                coef[r, 2] = np.float32(diffusion.alphas_cumprod_prev[i])
            tabs[key] = (tv.to(device), torch.from_numpy(coef).to(device), diffusion)     # (keeps the schedule object alive)
        return tabs[key][:2]

    @torch.no_grad()
    def ddim_sample_fused(self, noise: torch.Tensor, z: torch.Tensor, diffusion, cfg_scale: Optional[float],
                          per_token: Optional[torch.Tensor] = None) -> torch.Tensor:
        """GaussianDiffusion.ddim_sample_loop over forward_with_cfg (diffusion.py:714-794, dit.py:294-311) as ONE persistent
        launch (csrc/dit_fused.hip dit_sample_fused_k): noise [nb, T, A], z [N, 1, token] (N = 2 nb with guidance: [cond; uncond])
        -> the sample [nb, T, A].  What does not depend on x is prepared by three small launches: the z embedding and the
        timestep embeddings of the whole schedule.  ``per_token`` [N, P, per_token_size] (MemVLA): the perceptual keys / values of
        every block are projected once (precompute_per_kv) and the launch attends them in its per-attention phases."""
        from .... import kernels as K
        st = Fp32View(self.store)
        p, h = self.p, self.hidden_size
        N = z.shape[0]
        nb, T, A = noise.shape
        anchor = self._anchor()
        tv, coef = self._sampler_tables(diffusion, noise.device)
        ze = Fn.LinearFn.apply(z.reshape(N, self.token_size).float().contiguous(), anchor, st, p + "z_embedder.linear.weight",
                               p + "z_embedder.linear.bias", L.ACT_NONE, None)
        tf = K.timestep_embedding(tv, self._timestep_freqs(noise.device))
        te = Fn.MlpFn.apply(tf, anchor, st, p + "t_embedder.mlp.0.weight", p + "t_embedder.mlp.0.bias",
                            p + "t_embedder.mlp.2.weight", p + "t_embedder.mlp.2.bias", L.ACT_SILU)
        x = noise.float().contiguous().clone()
        self.used_fused = True
        self.store.wait_pending()                  # the kernel reads the masters through raw pointers
        if self.use_per_attn:
            assert per_token is not None and per_token.shape[0] == N
            K.dit_sample_bf16_fwd(x, ze.contiguous(), te.contiguous(), st.w(p + "positional_embedding").reshape(T + 1, h),
                                  st.w(p + "x_embedder.linear.weight"), st.w(p + "x_embedder.linear.bias"),
                                  st.w(p + "final_layer.linear.weight"), st.w(p + "final_layer.linear.bias"), coef, nb,
                                  cfg_scale is not None, float(cfg_scale or 0.0), self._packed_table(st), self.depth, T + 1, h,
                                  self.num_heads, self.mlp_hidden, 1e-6, per_kv=self.precompute_per_kv(per_token, packed=True))
            return x
        if self._bf16_sampler(N, T + 1):
            # a model served in bfloat16 multiplies with bf16 operands, like the reference's bf16 head (cogact_exp.py:134-138);
            # residual stream, LayerNorm, attention and accumulation stay fp32 (csrc/dit_fused.hip, dit_sample_bf16_k)
            K.dit_sample_bf16_fwd(x, ze.contiguous(), te.contiguous(), st.w(p + "positional_embedding").reshape(T + 1, h),
                                  st.w(p + "x_embedder.linear.weight"), st.w(p + "x_embedder.linear.bias"),
                                  st.w(p + "final_layer.linear.weight"), st.w(p + "final_layer.linear.bias"), coef, nb,
                                  cfg_scale is not None, float(cfg_scale or 0.0), self._packed_table(st), self.depth, T + 1, h,
                                  self.num_heads, self.mlp_hidden, 1e-6)
            return x
        K.dit_sample_fwd(x, ze.contiguous(), te.contiguous(), st.w(p + "positional_embedding").reshape(T + 1, h),
                         st.w(p + "x_embedder.linear.weight"), st.w(p + "x_embedder.linear.bias"),
                         st.w(p + "final_layer.linear.weight"), st.w(p + "final_layer.linear.bias"), coef, nb,
                         cfg_scale is not None, float(cfg_scale or 0.0), self._weight_table(st), self.depth, T + 1, h,
                         self.num_heads, self.mlp_hidden, 1e-6)
        return x

    def forward_with_cfg(self, x, t, z, cfg_scale=None, per_token=None, per_kv=None):
        """dit.py:294-311: both halves of the batch are the FIRST half of x; returns the RAW network output
        for [cond; uncond] — the guidance mix eps = u + s (c - u) is fused into dxa_ddim_step
        (diffusion.SpacedDiffusion.ddim_sample_loop receives cfg_scale)."""
        half = x[: len(x) // 2]
        return self.forward(torch.cat([half, half], dim=0), t, z, per_token=per_token, per_kv=per_kv)
