"""MemVLA policy: host-side mirror of dexbotic/model/memvla/memvla_arch.py on libdexbotic_amd kernels
(SURVEY.md §8a row A12).

= the CogACT path (vision tower, projector, splice, Qwen2 decoder, cognition token, DiT diffusion head) plus
``BottleneckSE`` perceptual compression (:129-167), ``PerCogMemBank`` (:190-409: per-episode banks of perceptual and
cognitive tokens, two ``CrossTransformerBlock`` retrieval layers over [bank ; timestep PE], ``GateFusion``, append +
token-merge consolidation) and a DiT whose blocks also cross-attend to the perceptual tokens.  The bank is STATEFUL and
order dependent exactly like the reference: the samples of a batch are walked in order, each retrieving from what the
earlier frames of its episode left behind (detached).  That loop is host logic; every tensor op in it is a kernel.

Retrieval dropout: the reference hard-codes ``dropout=0.1`` in its CrossTransformerBlocks (memvla_arch.py:83, 99-105,
120-123; no config key).  ``MemVLAConfig.retrieval_dropout`` defaults to that 0.1 and applies it while TRAINING (attention
weights inside the attention kernels + the two FFN dropouts); the deterministic goldens pass 0.0 explicitly.  Deviation: the
reference also hands dropout_p to SDPA in eval (stochastic inference); eval is deterministic here.
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import _lib as L
from ... import functional as Fn
from ... import hostcpu
from ... import kernels as K
from ...engine import ParamStore
from ..cogact.cogact_arch import CogActConfig, CogActModel, CogACTForCausalLM
from ..dexbotic_arch import CausalLMOutputDexbotic, register_with_hf

BANK = "model.per_cog_mem_bank."


class MemVLAConfig(CogActConfig):
    model_type = "dexbotic_memvla"

    def __init__(self, per_token_size: Optional[int] = None, dataloader_type: Optional[str] = None,
                 group_size: Optional[int] = None, mem_length: Optional[int] = None, retrieval_layers: Optional[int] = None,
                 use_timestep_pe: Optional[bool] = None, fusion_type: Optional[str] = None,
                 consolidate_type: Optional[str] = None, update_fused: bool = True, retrieval_dropout: float = 0.1,
                 **kwargs):
        super().__init__(**kwargs)
        self.per_token_size, self.dataloader_type, self.group_size = per_token_size, dataloader_type, group_size
        self.mem_length, self.retrieval_layers, self.use_timestep_pe = mem_length, retrieval_layers, use_timestep_pe
        self.fusion_type, self.consolidate_type, self.update_fused = fusion_type, consolidate_type, update_fused
        # dropout of the retrieval blocks (memvla_arch.py:83: the reference constructs them with dropout=0.1, not configurable):
        # 0.1 trains like the reference; 0.0 (default) is the deterministic setting the round-1 goldens pin
        self.retrieval_dropout = float(retrieval_dropout)


register_with_hf(MemVLAConfig)


def _lin(st, x, wn, bn, act=L.ACT_NONE, wshape=None):
    return Fn.LinearFn.apply(x, st.params[wn], st, wn, bn, act, wshape)


class BottleneckSE(nn.Module):
    """memvla_arch.py:129-167.  The 1x1 convolutions are token-wise linears ([out, in, 1, 1] weights used as [out, in])."""

    def __init__(self, store: ParamStore, prefix: str, C_in: int, C_out: int, reduction: int = 16, hidden_ratio: float = 0.5):
        super().__init__()
        self.store, self.p, self.C_in, self.C_out = store, prefix, C_in, C_out
        self.se, self.hm = max(1, C_in // reduction), max(1, int(C_in * hidden_ratio))
        store.new_bucket()
        store.register([(prefix + "excite.1.weight", (self.se, C_in, 1, 1)), (prefix + "excite.1.bias", (self.se,))])
        store.register([(prefix + "excite.3.weight", (C_in, self.se, 1, 1)), (prefix + "excite.3.bias", (C_in,))])
        store.register([(prefix + "reduce.0.weight", (self.hm, C_in, 1, 1)), (prefix + "reduce.0.bias", (self.hm,))])
        store.register([(prefix + "reduce.2.weight", (C_out, self.hm, 1, 1)), (prefix + "reduce.2.bias", (C_out,))])

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        st, p = self.store, self.p
        B, N, C_ = x.shape
        assert int(math.sqrt(N)) ** 2 == N, "Input feature has no spatial structure"
        m = Fn.TokenMeanFn.apply(x)                                                              # [B,C]
        g = _lin(st, m, p + "excite.1.weight", p + "excite.1.bias", L.ACT_RELU, (self.se, C_))
        g = _lin(st, g, p + "excite.3.weight", p + "excite.3.bias", L.ACT_SIGMOID, (C_, self.se))
        y = Fn.RowGateFn.apply(x, g).reshape(B * N, C_)
        y = _lin(st, y, p + "reduce.0.weight", p + "reduce.0.bias", L.ACT_RELU, (self.hm, C_))
        return _lin(st, y, p + "reduce.2.weight", p + "reduce.2.bias", L.ACT_NONE, (self.C_out, self.hm)).view(B, N, self.C_out)


class CrossTransformerBlock(nn.Module):
    """memvla_arch.py:84-127: post-LN cross attention, 4 heads, GELU(erf) FFN.  ``dropout`` p > 0: SDPA's dropout on the
    attention weights (:120-123) and the two nn.Dropout of the FFN (:99-105), applied while training; the masks come from
    ``mask_fn(shape) -> tensor of 0 | 1/(1-p)`` (default: a device draw), in the order attention, FFN hidden, FFN output.
    The reference also hands dropout_p to SDPA in eval (stochastic inference); here eval is deterministic."""

    def __init__(self, store: ParamStore, prefix: str, feature_dim: int, num_heads: int = 4, dropout: float = 0.0):
        super().__init__()
        assert feature_dim % num_heads == 0, "feature_dim % num_heads must be 0"
        self.store, self.p, self.D, self.H = store, prefix, feature_dim, num_heads
        self.dropout = float(dropout)
        self.mask_fn = None
        D = feature_dim
        for n in ("q_proj", "k_proj", "v_proj"):
            store.register([(prefix + n + ".weight", (D, D)), (prefix + n + ".bias", (D,))])
        store.register([(prefix + "attn_norm.weight", (D,)), (prefix + "attn_norm.bias", (D,))], layernorm=True)
        store.register([(prefix + "ffn.0.weight", (4 * D, D)), (prefix + "ffn.0.bias", (4 * D,))])
        store.register([(prefix + "ffn.3.weight", (D, 4 * D)), (prefix + "ffn.3.bias", (D,))])
        store.register([(prefix + "ffn_norm.weight", (D,)), (prefix + "ffn_norm.bias", (D,))], layernorm=True)

    def _mask(self, shape, like: torch.Tensor) -> torch.Tensor:
        if self.mask_fn is not None:
            m = self.mask_fn(tuple(shape))
            return torch.as_tensor(m).to(device=like.device, dtype=like.dtype).reshape(shape).contiguous()
        return K.dropout_mask(shape, self.dropout, like.dtype, like.device)      # one launch (dxa_dropout_mask)

    def forward(self, query: torch.Tensor, k: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
        st, p, D, H = self.store, self.p, self.D, self.H
        B, N, _ = query.shape
        M = k.shape[1]
        hd = D // H
        drop = self.dropout > 0.0 and self.training
        # tensors with two consumers are forked explicitly (Fn.ForkFn): their gradient sums are then library launches
        q2, q2r = Fn.ForkFn.apply(query.reshape(B * N, D))
        qp = _lin(st, q2, p + "q_proj.weight", p + "q_proj.bias").view(B, N, H, hd)
        kp = _lin(st, k.reshape(B * M, D), p + "k_proj.weight", p + "k_proj.bias").view(B, M, H, hd)
        vp = _lin(st, v.reshape(B * M, D), p + "v_proj.weight", p + "v_proj.bias").view(B, M, H, hd)
        o = Fn.AttnFn.apply(qp, kp, vp, self._mask((B, H, N, M), qp) if drop else None).reshape(B * N, D)
        anchor = st.params[p + "attn_norm.weight"]
        x = Fn.NormFn.apply(Fn.AddFn.apply(q2r, o), anchor, st, "ln", p + "attn_norm.weight", p + "attn_norm.bias", 1e-5)
        x, xr = Fn.ForkFn.apply(x)
        if drop:
            h = _lin(st, x, p + "ffn.0.weight", p + "ffn.0.bias", L.ACT_GELU_ERF)
            h = Fn.DropFn.apply(h, self._mask((B * N, 4 * D), h))
            f = _lin(st, h, p + "ffn.3.weight", p + "ffn.3.bias")
            f = Fn.DropFn.apply(f, self._mask((B * N, D), f))
        else:
            f = Fn.MlpFn.apply(x, anchor, st, p + "ffn.0.weight", p + "ffn.0.bias", p + "ffn.3.weight", p + "ffn.3.bias",
                               L.ACT_GELU_ERF)
        y = Fn.NormFn.apply(Fn.AddFn.apply(xr, f), anchor, st, "ln", p + "ffn_norm.weight", p + "ffn_norm.bias", 1e-5)
        return y.view(B, N, D)


class _BankBuf:
    """one episode's memory of one role: entries oldest first in one device buffer, timesteps beside them, count on the host"""
    __slots__ = ("feat", "ts", "sims", "n")

    def __init__(self, cap: int, N: int, D: int, dtype, device):
        self.feat = torch.empty((cap, N, D), device=device, dtype=dtype)
        self.ts = torch.empty(cap, device=device, dtype=torch.float32)
        self.sims = torch.empty(max(cap, 2), device=device, dtype=torch.float32)
        self.n = 0


class PerCogMemBank(nn.Module):
    """memvla_arch.py:190-444.  'group' training batches (banks cleared per batch, :330-333) and the single-episode
    eval path; 'gate' fusion; 'tome' (token merge) or 'fifo' consolidation; timestep PE."""

    def __init__(self, store: ParamStore, dataloader_type: str, group_size: int, per_token_size: int, cog_token_size: int,
                 mem_length: int = 16, retrieval_layers: int = 2, use_timestep_pe: bool = True, fusion_type: str = "gate",
                 consolidate_type: str = "tome", update_fused: bool = True, retrieval_dropout: float = 0.0):
        super().__init__()
        assert dataloader_type in ("stream", "group", "parallel_stream")
        assert fusion_type in ("gate", "add") and consolidate_type in ("fifo", "tome")
        if dataloader_type != "group":
            raise NotImplementedError("native PerCogMemBank: the 'group' dataloader mode (the reference trainer's sampler)")
        if not use_timestep_pe or fusion_type != "gate":
            raise NotImplementedError("native PerCogMemBank: timestep PE + gate fusion (the reference defaults)")
        self.store = store
        self.roles = ("per", "cog")
        self.dataloader_type, self.group_size, self.mem_length = dataloader_type, group_size, mem_length
        self.retrieval_layers, self.consolidate_type, self.update_fused = retrieval_layers, consolidate_type, update_fused
        self.token_dim = {"per": per_token_size, "cog": cog_token_size}
        store.new_bucket()
        self.blocks = {r: [CrossTransformerBlock(store, f"{BANK}retrieval_blocks.{r}.{i}.", self.token_dim[r],
                                                 dropout=retrieval_dropout)
                           for i in range(retrieval_layers)] for r in self.roles}
        for r in self.roles:
            D = self.token_dim[r]
            store.register([(f"{BANK}gate_fusion_blocks.{r}.proj.weight", (D, 2 * D)),
                            (f"{BANK}gate_fusion_blocks.{r}.proj.bias", (D,))])
        for r in self.roles:
            D = self.token_dim[r]
            store.register([(f"{BANK}timestep_embedders.{r}.mlp.0.weight", (D, 256)),
                            (f"{BANK}timestep_embedders.{r}.mlp.0.bias", (D,))])
            store.register([(f"{BANK}timestep_embedders.{r}.mlp.2.weight", (D, D)),
                            (f"{BANK}timestep_embedders.{r}.mlp.2.bias", (D,))])
        self._freqs = {}
        self.reset()

    def reset(self):
        # banks[role][episode] = _BankBuf: the entries (oldest first) live side by side in ONE device buffer [mem_length + 1, N, D]
        # with their timesteps [mem_length + 1] (fp32, device: after a token merge a timestep is the mean of two, chosen on the
        # device); the host only keeps the COUNT, which evolves without reading anything back
        self.banks: Dict[str, Dict[tuple, "_BankBuf"]] = {r: {} for r in self.roles}

    def entries(self, role: str, eid=(0, 0)) -> List[Tuple[torch.Tensor, torch.Tensor]]:
        """[(timestep, feat[N, D])] of one bank, oldest first (tests / inspection; the reference's list layout)"""
        b = self.banks[role].get(tuple(eid))
        return [] if b is None else [(b.ts[i], b.feat[i]) for i in range(b.n)]

    def train(self, mode: bool = True):
        super().train(mode)
        for blks in self.blocks.values():           # held in a plain dict (the parameters live in the arena)
            for b in blks:
                b.train(mode)
        return self

    def set_mask_fn(self, fn) -> None:
        """tests: ``fn(shape) -> array of 0 | 1/(1-p)`` replaces the device draw of every dropout mask of the retrieval blocks
        (called in the order the blocks run: per sample, per role, per layer: attention, FFN hidden, FFN output)"""
        for blks in self.blocks.values():
            for b in blks:
                b.mask_fn = fn

    # ---- pieces -------------------------------------------------------------------------------------------
    def _encode_time(self, role: str, t: torch.Tensor, dtype) -> torch.Tensor:
        """TimestepEmbedder (:36-81): sinusoid(256) -> Linear -> SiLU -> Linear, in the compute dtype"""
        st = self.store
        key = str(t.device)
        if key not in self._freqs:
            self._freqs[key] = torch.exp(-math.log(10000) * torch.arange(0, 128, dtype=torch.float32) / 128).to(t.device)
        p = f"{BANK}timestep_embedders.{role}."
        e = K.timestep_embedding(t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous(), self._freqs[key])
        if dtype != torch.float32:
            e = K.cast(e, dtype)
        return Fn.MlpFn.apply(e, st.params[p + "mlp.0.weight"], st, p + "mlp.0.weight", p + "mlp.0.bias",
                              p + "mlp.2.weight", p + "mlp.2.bias", L.ACT_SILU)

    @torch.no_grad()
    def _consolidate(self, role: str, eid, feat: torch.Tensor, timestep: torch.Tensor) -> None:
        """append (detached copy) + consolidation (_memory_consolidate, memvla_arch.py:289-306): library copies into the bank's
        buffer, then — once the bank is over mem_length — ONE device-side merge (dxa_bank_consolidate: similarities, arg-max,
        merge, compaction; 'fifo': drop of the oldest) per surplus entry.  Nothing is read back."""
        N, D = feat.shape[-2] if feat.dim() > 1 else 1, feat.shape[-1]
        bank = self.banks[role].get(eid)
        if bank is None:
            bank = self.banks[role][eid] = _BankBuf(self.mem_length + 1, N, D, feat.dtype, feat.device)
        assert bank.n <= self.mem_length
        K.cast(feat.detach().reshape(1, N, D).contiguous(), bank.feat.dtype, out=bank.feat[bank.n:bank.n + 1])
        K.cast(timestep.reshape(1), torch.float32, out=bank.ts[bank.n:bank.n + 1])
        bank.n += 1
        while bank.n > self.mem_length:
            K.bank_consolidate(bank.feat, bank.ts, bank.n, bank.sims, fifo=self.consolidate_type == "fifo")
            bank.n -= 1

    def _process_batch(self, role: str, tokens: torch.Tensor, episode_ids, timesteps: torch.Tensor) -> torch.Tensor:
        """``timesteps``: ONE fp32 device tensor [B] (a single upload per batch)"""
        st = self.store
        B, N, D = tokens.shape
        if self.training:
            self.banks[role].clear()                                   # 'group' (:330-333)
        else:
            episode_ids = [(0, 0) for _ in range(B)]
        gp = f"{BANK}gate_fusion_blocks.{role}."
        nl = len(self.blocks[role])
        grad = torch.is_grad_enabled() and tokens.requires_grad
        rows = Fn.UnbindRowsFn.apply(tokens) if B > 1 else (tokens,)
        outs = []
        for i in range(B):
            eid = tuple(episode_ids[i])
            bank = self.banks[role].get(eid)
            has_hist = bank is not None and bank.n > 0
            # consumers of this sample's tokens: the query of block 0, the two gate-fusion operands (+ without history: the key
            # sum and one value per block, the frame retrieving from itself, :345-349)
            fork = Fn.ForkFn.apply(rows[i], 3 if has_hist else 4 + nl)
            working, w_cat, w_fuse = fork[0], fork[1].reshape(N, D), fork[2].reshape(N, D)
            if has_hist:
                T_ = bank.n
                # a snapshot: the bank's buffer is overwritten by the next frames while this sample's backward still needs
                # what it retrieved from (k / v projection inputs)
                mem = K.cast(bank.feat[:T_], bank.feat.dtype) if grad else bank.feat[:T_]
                pe = self._encode_time(role, bank.ts[:T_], tokens.dtype)          # (T,D)
                vals = [mem.reshape(1, T_ * N, D)] * nl
            else:
                T_ = 1
                mem = fork[3]
                pe = self._encode_time(role, timesteps[i:i + 1], tokens.dtype)
                vals = [fork[4 + l] for l in range(nl)]
            key = Fn.AddRowsFn.apply(mem.reshape(T_, N, D), pe).reshape(1, T_ * N, D)
            keys = Fn.ForkFn.apply(key, nl) if nl > 1 else (key,)
            q = working
            for l, blk in enumerate(self.blocks[role]):
                q = blk(q, keys[l], vals[l])
            q_cat, q_fuse = Fn.ForkFn.apply(q.reshape(N, D))
            scale = _lin(st, Fn.CatLastFn.apply(w_cat, q_cat), gp + "proj.weight", gp + "proj.bias", L.ACT_SIGMOID)
            fused = Fn.GateFuseFn.apply(scale, w_fuse, q_fuse).view(1, N, D)
            outs.append(fused)
            self._consolidate(role, eid, fused[0] if self.update_fused else tokens[i], timesteps[i:i + 1])
        return Fn.CatRowsFn.apply(*outs) if B > 1 else outs[0]

    def process_batch_per(self, per_tokens, episode_ids, timesteps):
        return self._process_batch("per", per_tokens, episode_ids, timesteps)

    def process_batch_cog(self, cog_tokens, episode_ids, timesteps):
        return self._process_batch("cog", cog_tokens, episode_ids, timesteps)


class MemVLAModel(CogActModel):
    def __init__(self, config: MemVLAConfig, store: ParamStore):
        # same registration order as the forward: tower, projector, decoder (CogActModel) then the memory modules;
        # the action head is built last by CogActModel.__init__ through build_action_model(config with per_token_size)
        action_type = config.action_model_type
        config.action_model_type = None
        CogActModel.__init__(self, config, store)
        config.action_model_type = action_type
        self.per_compr = None
        self.per_cog_mem_bank = None
        if getattr(config, "per_token_size", None) is not None:
            self.per_compr = BottleneckSE(store, "model.per_compr.", config.hidden_size, config.per_token_size)
        need = ["dataloader_type", "group_size", "mem_length", "retrieval_layers", "use_timestep_pe", "fusion_type",
                "consolidate_type", "per_token_size"]
        if all(getattr(config, k, None) is not None for k in need):
            self.per_cog_mem_bank = PerCogMemBank(
                store, config.dataloader_type, config.group_size, config.per_token_size, config.hidden_size,
                config.mem_length, config.retrieval_layers, config.use_timestep_pe, config.fusion_type,
                config.consolidate_type, getattr(config, "update_fused", True),
                retrieval_dropout=float(getattr(config, "retrieval_dropout", 0.1) or 0.0))
        if action_type is not None:
            self.action_head = self._build_action_head_module(config)


class MemVLAForCausalLM(CogACTForCausalLM):
    config_class = MemVLAConfig
    coalescible_micro_batches = False      # the memory bank walks the batch in order: micro-batches are not interchangeable
    gradient_side_stream = True            # round 6: 310.7 -> 307.8 ms per step now that the step is GPU-bound (profiles/r06_memvla_hostbound.txt);
                                           # while its 9,000 launches were host-bound it measured 363 -> 372 (round 4) and was off

    def _real_init(self, config: MemVLAConfig):
        self.model = MemVLAModel(config, self.store)
        self.store.new_bucket()
        self.store.register([("lm_head.weight", (config.vocab_size, config.hidden_size))])
        self.cur_timestep = 0          # inference

    def unused_parameter_names(self) -> List[str]:
        names = ["lm_head.weight"] + self.model.mm_vision_tower.unused_parameter_names()
        return names

    def train(self, mode: bool = True):
        super().train(mode)
        if self.model.per_cog_mem_bank is not None:
            self.model.per_cog_mem_bank.train(mode)
        return self

    def _vlm(self, input_ids, attention_mask, images):
        """prefill -> (last hidden state [B,S,d], projector output = vision_proj_feats [B, V*N_v, d])"""
        feats = {}

        def hook(_, __, out):
            feats["vision_proj"] = out
        h = self.model.mm_projector.register_forward_hook(hook)
        try:
            (_, _, attention_mask, _, inputs_embeds, _, _) = self.model._prepare_inputs_labels_for_multimodal(
                input_ids, None, attention_mask, None, None, None, images)
        finally:
            h.remove()
        return self.model.run_llm(inputs_embeds, attention_mask), feats["vision_proj"], attention_mask

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, inputs_embeds=None,
                labels=None, use_cache=None, output_attentions=None, output_hidden_states=None, images=None,
                return_dict=None, cache_position=None, actions=None, states=None, repeated_diffusion_steps: int = 4,
                indexes=None, **kwargs) -> CausalLMOutputDexbotic:
        """memvla_arch.py:546-664.  kwargs ``noise`` / ``timesteps`` / ``drop_ids`` inject the diffusion draws."""
        hidden, vision_proj, attention_mask = self._vlm(input_ids, attention_mask, images)
        B, S, d = hidden.shape
        loss = None
        if attention_mask is not None and actions is not None:
            plan = self.model._last_plan
            pageable = os.environ.get("DXA_MEMVLA_PAGEABLE_UPLOAD", "0") == "1"       # (measurement switch: the round-5 uploads)
            idx = plan.dev(hidden.device)["last_flat"] if not pageable else \
                torch.from_numpy(np.arange(B, dtype=np.int64) * S + plan.last_index).to(hidden.device)
            cog = Fn.GatherRowsFn.apply(hidden.reshape(B * S, d), idx).to(hidden.dtype).view(B, 1, d)
            per = self.model.per_compr(vision_proj.reshape(B, -1, d))
            eids = [tuple(int(v) for v in item[:2]) for item in indexes]
            # from PINNED memory: a pageable source makes the copy wait for the stream to drain — here, at the end of the decoder's
            # forward, that throws away the 60 ms the launching thread is ahead and leaves the bank's ~2,400 small launches
            # host-bound (profiles/r06_host_uploads.txt)
            ts = torch.tensor([float(item[2]) for item in indexes], dtype=torch.float32)
            ts = ts.to(hidden.device, non_blocking=True) if pageable else hostcpu.upload(ts, hidden.device)
            bank = self.model.per_cog_mem_bank
            cog = bank.process_batch_cog(cog, eids, ts)
            per = bank.process_batch_per(per, eids, ts)
            A, T = self.config.action_dim, self.config.chunk_size
            acts = actions.reshape(actions.size(0), -1, A).float()[:, :T, :]
            R = repeated_diffusion_steps
            # the reference repeats the perceptual tokens with the actions (:515-519) and projects all R copies in every DiT block;
            # the head takes the B distinct sequences and the repeat count instead (DiT.forward per_repeat: same values, the
            # embedding and the 24 key/value projections on B x P rows instead of R x B x P).  DXA_MEMVLA_PER_REPEAT=1: as written there
            if os.environ.get("DXA_MEMVLA_PER_REPEAT", "0") == "1":
                per_kw = dict(per_token=per.float().repeat(R, 1, 1))
            else:
                per_kw = dict(per_token=per.float(), per_repeat=R)
            loss = self.model.action_head_module.loss(
                acts.repeat(R, 1, 1), cog.float().repeat(R, 1, 1), noise=kwargs.get("noise"), timestep=kwargs.get("timesteps"),
                drop_ids=kwargs.get("drop_ids"), **per_kw)
        return CausalLMOutputDexbotic(loss=loss, logits=hidden, hidden_states=(hidden,))

    @torch.no_grad()
    def inference_action(self, input_ids, image_tensor, episode_first_frame, inference_args={}, **kwargs):
        """memvla_arch.py:666-746: one frame of a running episode ('True' on the first frame resets the memory)"""
        cfg_scale = inference_args.get("cfg_scale", 1.5)
        num_ddim_steps = inference_args.get("num_ddim_steps", 10)
        action_norms = inference_args.get("action_norms")
        assert episode_first_frame in ("True", "False"), "episode_first_frame must be 'True' or 'False'"
        bank = self.model.per_cog_mem_bank
        if episode_first_frame == "True":
            bank.reset()
            self.cur_timestep = 0
        dev = self.store.device
        hidden, vision_proj, _ = self._vlm(input_ids, None, image_tensor.to(device=dev, dtype=self.store.compute_dtype))
        B, d = hidden.shape[0], hidden.shape[-1]
        cog = hidden[:, -1, :].unsqueeze(1).contiguous()
        per = self.model.per_compr(vision_proj.reshape(B, -1, d))
        ts = hostcpu.upload(torch.tensor([float(self.cur_timestep)], dtype=torch.float32), dev)    # (pinned: no wait for the prefill)
        self.cur_timestep += 1
        cog = bank.process_batch_cog(cog, [(0, 0)], ts)
        per = bank.process_batch_per(per, [(0, 0)], ts)
        head = self.model.action_head
        noise = kwargs.get("noise")
        if noise is None:
            noise = torch.randn(B, self.config.chunk_size, self.config.action_dim, device=dev, dtype=torch.float32)
        noise = noise.to(device=dev, dtype=torch.float32)
        if head.ddim_diffusion is None or head.ddim_diffusion.num_timesteps != num_ddim_steps:
            head.create_ddim(ddim_step=num_ddim_steps)
        cogf = cog.float()
        if cfg_scale > 1.0:
            noise = torch.cat([noise, noise], 0)
            unc = self.store.w32("model.action_head.net.z_embedder.uncondition")
            z = torch.cat([cogf, unc.unsqueeze(0).expand(B, 1, -1)], 0)
            model_kwargs = dict(z=z, cfg_scale=cfg_scale)
            sample_fn = head.net.forward_with_cfg
        else:
            model_kwargs = dict(z=cogf)
            sample_fn = head.net.forward
        per_token = per.float().repeat(2, 1, 1) if cfg_scale > 1.0 else per.float()
        cfg = model_kwargs.get("cfg_scale")

        fused = inference_args.get("fused_sampler", True) and inference_args.get("cache_per_kv", True) and \
            head.net.fused_sampler_ok(noise.shape[0], self.config.chunk_size + 1, per_token.shape[1])
        head.net.used_fused = bool(fused)

        def sample(noise, z, per_token):
            if fused:
                # bf16 serving: all DDIM steps of the DiT with perceptual attention in ONE persistent launch (csrc/dit_fused.hip)
                return head.net.ddim_sample_fused(noise[:B], z, head.ddim_diffusion, cfg, per_token=per_token)     # [B, T, A]
            # the perceptual keys/values of the 24 blocks do not depend on the DDIM step: projected once per request
            mk = dict(z=z, per_kv=head.net.precompute_per_kv(per_token)) if inference_args.get("cache_per_kv", True) \
                else dict(z=z, per_token=per_token)
            if cfg is not None:
                mk["cfg_scale"] = cfg
            return head.ddim_diffusion.ddim_sample_loop(sample_fn, noise.shape, noise, clip_denoised=False, model_kwargs=mk,
                                                        eta=0.0, device=dev)
        from ... import graphs
        if dev.type == "cuda" and inference_args.get("use_graph", graphs.enabled()):
            # the sampler (DiT-L with perceptual attention, ~480 launches per DDIM step) is host-bound: captured once per
            # shape into a HIP graph and replayed (graphs.GraphCache); the stateful memory bank above stays eager host logic
            cache = self.__dict__.setdefault("_sampler_graphs", graphs.GraphCache(dev))
            if fused:
                head.net.refresh_packed()          # the bf16 operand copy a captured launch reads: up to date before a replay
            run_stream = cache
            samples = cache.run(("ddim", float(cfg_scale), int(num_ddim_steps), bool(fused)), sample,
                                dict(noise=noise, z=model_kwargs["z"].contiguous(), per_token=per_token.contiguous()))
        else:
            run_stream = None
            samples = sample(noise, model_kwargs["z"], per_token)
        if cfg_scale > 1.0:
            samples = samples[:B]
        host = samples[0].cpu().numpy()
        if fused and K.dit_blocks_timed_out(run_stream.stream if run_stream is not None else None):
            # the persistent launch gave up at a barrier (its workgroups were not co-resident: something else held CUs for seconds):
            # the frame is sampled again block by block, which this process keeps doing from now on.  The memory bank above was
            # updated once and is not touched again.
            head.net.allow_fused = False
            self.__dict__.pop("_sampler_graphs", None)
            fused = False
            samples = sample(noise, model_kwargs["z"], per_token)
            host = (samples[:B] if cfg_scale > 1.0 else samples)[0].cpu().numpy()
        return self._denorm(host, action_norms).tolist()


from ..dexbotic_arch import register_model_with_hf  # noqa: E402

register_model_with_hf(MemVLAForCausalLM)
