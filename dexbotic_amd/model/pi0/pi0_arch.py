"""pi0 policy (SigLIP + dual-expert Gemma mixture of transformers + flow-matching action head): host-side mirror of
dexbotic/model/pi0/pi0_arch.py on libdexbotic_amd kernels.

``Pi0Config`` (:53-83), ``Pi0Model`` (:86-106: llm + action_expert + the five small linears), ``Pi0ForCausalLM``:
``embed_prefix`` (:223-259), ``embed_suffix`` (:261-315), ``_inner_forward_mot`` (:116-216) and
``inference_action`` (:402-491: prefix pass fills a K/V cache, 10 Euler steps x += v dt re-encode the suffix against
it).  What differs is underneath: parameters live in the flat arenas of engine.ParamStore; every layer is two GEMM
halves per expert around ONE attention call over both experts' tokens, with the reference's block mask expressed as
per-query key counts + per-key validity (cumsum(ar_mask) is non-decreasing, so "cumsum[j] <= cumsum[i]" is a prefix).

Training: ``forward`` builds the flow-matching loss through ``functional.Pi0MotLayerFn`` (one autograd node per
layer over both experts, gradients written straight into the arena); inference: ``inference_action``.
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np
import torch
import torch.nn as nn

from ... import _lib as L
from ... import functional as Fn
from ... import hostcpu
from ... import kernels as K
from ...engine import ParamStore
from ..dexbotic_arch import (ActionOutputForCausalLM, CausalLMOutputDexbotic, NativePreTrainedMixin, _DTYPES, _hf_config_base,
                             register_with_hf)
from ..llm.gemma import GemmaConfig, GemmaExpert
from ..modules.mm_projector.builder import build_vision_projector
from ..modules.mm_vision.builder import build_vision_tower
from ..modules.mm_vision.siglip.siglip_encoder import SiglipVisionConfig


class Pi0Config(_hf_config_base()):
    """``transformers.PretrainedConfig`` subclass registered with ``AutoConfig`` under the reference's ``model_type``
    (pi0_arch.py:54-110): ``AutoConfig.from_pretrained`` on a reference pi0 checkpoint directory resolves to it, the nested
    vision / llm / action-expert configs round-trip through ``config.json``."""
    model_type = "dexbotic_pi0"

    def __init__(self, vision_config=None, processor_config=None, action_config=None, llm_config=None,
                 mm_projector_type: str = "linear", action_dim: int = 32, chunk_size: int = 50,
                 compute_dtype="float32", **kwargs):
        self.vision_config = SiglipVisionConfig.from_any(vision_config if vision_config is not None else {})
        self.processor_config = processor_config
        self.action_config = GemmaConfig.from_any(action_config if action_config is not None else {})
        self.llm_config = GemmaConfig.from_any(llm_config if llm_config is not None else {})
        self.mm_projector_type = mm_projector_type
        self.action_dim, self.chunk_size = action_dim, chunk_size
        self.compute_dtype = compute_dtype if isinstance(compute_dtype, str) else str(compute_dtype).replace("torch.", "")
        a, l_ = self.action_config, self.llm_config
        if (a.num_hidden_layers, a.num_attention_heads, a.num_key_value_heads, a.head_dim) != (
                l_.num_hidden_layers, l_.num_attention_heads, l_.num_key_value_heads, l_.head_dim):
            raise ValueError("the two experts share one attention: depth, heads and head_dim must match")
        for k in ("model_type", "architectures", "transformers_version"):
            kwargs.pop(k, None)
        hidden, vocab = kwargs.pop("hidden_size", None), kwargs.pop("vocab_size", None)
        super().__init__(**kwargs)
        self.hidden_size = self.llm_config.hidden_size if hidden is None else hidden
        self.vocab_size = self.llm_config.vocab_size if vocab is None else vocab

    def to_dict(self):
        out = {}
        for k, v in super().to_dict().items():
            if hasattr(v, "to_dict"):
                v = v.to_dict()
            elif isinstance(v, torch.dtype):
                v = str(v).replace("torch.", "")
            out[k] = v
        out["model_type"] = self.model_type
        return out

    def to_diff_dict(self):
        return self.to_dict()

    @classmethod
    def from_dict(cls, d, **kwargs):
        d = dict(d)
        for k in ("model_type", "architectures", "transformers_version"):
            d.pop(k, None)
        config = cls(**d)
        if kwargs.pop("return_unused_kwargs", False):
            unused = {k: v for k, v in kwargs.items() if not k.startswith("_") and k not in ("name_or_path", "trust_remote_code")}
            return config, unused
        return config

    @classmethod
    def from_pretrained(cls, path: str, **kwargs):
        import json
        import os
        with open(os.path.join(path, "config.json")) as f:
            return cls.from_dict(json.load(f), **kwargs)

    def save_pretrained(self, path: str, **kwargs) -> None:
        import json
        import os
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, "config.json"), "w") as f:
            json.dump(self.to_dict(), f, indent=2, default=str)


register_with_hf(Pi0Config)


class Pi0Model(nn.Module):
    """registration order = forward order: vision tower, projector, llm expert, action expert, suffix/head linears"""

    def __init__(self, config: Pi0Config, store: ParamStore):
        super().__init__()
        self.config, self.store = config, store
        self.mm_vision_tower = build_vision_tower(config.vision_config, store, "model.mm_vision_tower.",
                                                  processor_config=config.processor_config, select_layer=None)
        config.mm_hidden_size = self.mm_vision_tower.hidden_size
        self.mm_projector = build_vision_projector(config, store, "model.mm_projector.")
        self.llm = GemmaExpert(store, "model.llm.", config.llm_config)
        self.action_expert = GemmaExpert(store, "model.action_expert.", config.action_config)
        da, A = config.action_config.hidden_size, config.action_dim
        store.new_bucket()
        for name, shape in (("state_proj", (da, A)), ("action_in_proj", (da, A)), ("action_time_mlp_in", (da, 2 * da)),
                            ("action_time_mlp_out", (da, da)), ("action_out_proj", (A, da))):
            store.register([(f"model.{name}.weight", shape), (f"model.{name}.bias", (shape[0],))])

    @property
    def backbone(self):
        return self.llm

    @property
    def mm_vision_module(self):
        return self.mm_vision_tower

    @property
    def mm_projector_module(self):
        return self.mm_projector


def posemb_sincos(time: np.ndarray, dim: int, min_period: float = 4e-3, max_period: float = 4.0) -> np.ndarray:
    """pi0_arch.py:36-51 on the host (the schedule is known before any device work): float64 periods, the time in
    float32, sin/cos of the float64 quotient — returned float64 like the reference tensor"""
    frac = np.linspace(0.0, 1.0, dim // 2, dtype=np.float64)
    period = min_period * (max_period / min_period) ** frac
    x = time.astype(np.float32)[:, None].astype(np.float64) / period[None, :] * 2 * np.pi
    return np.concatenate([np.sin(x), np.cos(x)], axis=-1)


class Pi0ForCausalLM(NativePreTrainedMixin, nn.Module, ActionOutputForCausalLM):
    config_class = Pi0Config
    # trainer.NativeTrainer: the bias gradients' column sums (and any fp32 dW product) beside the dX chain on a side stream:
    # 250.0 -> 245.3 ms per step, two alternating runs each in one box (profiles/r06_memvla_hostbound.txt)
    gradient_side_stream = True

    def __init__(self, config: Pi0Config, device=None, train: bool = True):
        super().__init__()
        self.config = config
        device = device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu")
        self.store = ParamStore(device, _DTYPES[config.compute_dtype])
        self.model = Pi0Model(config, self.store)
        self._finish_init(train)

    @property
    def device(self):
        return self.store.device

    @property
    def dtype(self):
        return self.store.compute_dtype

    def unused_parameter_names(self) -> List[str]:
        """no gradient on the pi0 loss: SigLIP pooling head, the action expert's token embedding, and what only
        feeds prefix_out (last llm layer after its K/V projections, llm final norm)"""
        st, c = self.store, self.config
        last = f"model.llm.layers.{c.llm_config.num_hidden_layers - 1}."
        names = self.model.mm_vision_tower.unused_parameter_names()
        names += ["model.action_expert.embed_tokens.weight", "model.llm.norm.weight"]
        names += [n for n in st.slots if n.startswith(last) and
                  (".o_proj." in n or ".mlp." in n or "post_attention_layernorm" in n)]
        return names

    # ------------------------------------------------------------------------------------ embeddings
    def encode_images(self, images: torch.Tensor) -> torch.Tensor:
        return self.model.mm_projector(self.model.mm_vision_tower(images))

    @staticmethod
    def _to_dev(a: np.ndarray, device) -> torch.Tensor:
        """small host array -> device through pinned memory, non-blocking (hostcpu.upload: a pageable source would make the copy wait
        for the stream to drain)"""
        return hostcpu.upload(a, device)

    def prefix_mask(self, attention_mask, image_masks) -> np.ndarray:
        """input_mask np.bool [B, CAM * T + L] of embed_prefix from the two host-side masks alone (T = tokens per camera): the
        training step computes every mask and position of the mixture BEFORE it launches the vision tower — fetching a device
        mask with .cpu() afterwards waits for the tower, and the 7 ms of numpy that follow run with the GPU idle"""
        T = self.model.mm_vision_tower.num_patches
        im = np.asarray(image_masks.cpu() if torch.is_tensor(image_masks) else image_masks, dtype=bool)
        am = np.asarray(attention_mask.cpu() if torch.is_tensor(attention_mask) else attention_mask, dtype=bool)
        return np.concatenate([np.repeat(im, T, axis=1), am], axis=1)

    def embed_prefix(self, input_ids, attention_mask, images, image_masks, input_mask: Optional[np.ndarray] = None):
        """-> tokens [B, P, d] (compute dtype), input_mask np.bool [B, P], ar_mask np.bool [P] (all False).
        ``input_mask``: prefix_mask(...) computed by the caller beforehand."""
        B, CAM = images.shape[:2]
        dev, cdt = self.store.device, self.store.compute_dtype
        # all cameras in one tower pass: [B, CAM, ...] -> camera-major tokens like the reference's per-camera loop
        feats = self.encode_images(images.to(device=dev).transpose(0, 1).reshape(B * CAM, *images.shape[2:]))
        T = feats.shape[1]
        img_tok = feats.view(CAM, B, T, -1).permute(1, 0, 2, 3).reshape(B, CAM * T, -1)
        txt = self.model.llm.embed(input_ids.to(dev))
        tokens = torch.cat([img_tok.to(cdt), txt.to(cdt)], dim=1)
        if input_mask is None:
            im = np.asarray(image_masks.cpu() if torch.is_tensor(image_masks) else image_masks, dtype=bool)
            am = np.asarray(attention_mask.cpu() if torch.is_tensor(attention_mask) else attention_mask, dtype=bool)
            input_mask = np.concatenate([np.repeat(im, T, axis=1), am], axis=1)
        assert input_mask.shape[1] == tokens.shape[1], (input_mask.shape, tokens.shape)
        return tokens, input_mask, np.zeros(tokens.shape[1], dtype=bool)

    def embed_suffix(self, states: torch.Tensor, noisy_actions: torch.Tensor, time: Optional[np.ndarray], te: Optional[torch.Tensor] = None):
        """-> tokens [B, 1 + chunk, d_a] fp32->compute dtype, input_mask (all True), ar_mask [True, True, False...].
        ``te``: the sin/cos time embedding [B, d_a] already on the device (the sampler precomputes its schedule)."""
        st, c = self.store, self.config
        cdt = st.compute_dtype
        B, da = states.shape[0], c.action_config.hidden_size
        lin = lambda x, n, act=L.ACT_NONE: Fn.LinearFn.apply(x, st.params[f"model.{n}.weight"], st, f"model.{n}.weight",
                                                             f"model.{n}.bias", act, None)
        state_tok = lin(states.to(cdt), "state_proj").view(B, 1, da)
        if te is None:
            te = self._to_dev(posemb_sincos(time, da), st.device).to(cdt)                       # [B, da]
        act_tok = lin(noisy_actions.to(cdt).reshape(B * c.chunk_size, -1), "action_in_proj").view(B, c.chunk_size, da)
        h = torch.cat([act_tok, te[:, None, :].expand(B, c.chunk_size, da)], dim=-1).reshape(B * c.chunk_size, 2 * da)
        h = lin(h.contiguous(), "action_time_mlp_in", L.ACT_SILU)
        h = lin(h, "action_time_mlp_out").view(B, c.chunk_size, da)
        tokens = torch.cat([state_tok, h], dim=1)
        mask = np.ones((B, 1 + c.chunk_size), dtype=bool)
        ar = np.array([True, True] + [False] * (c.chunk_size - 1))
        return tokens, mask, ar

    # ------------------------------------------------------------------------------ mixture forward
    @staticmethod
    def _mask_tensors(q_cum: np.ndarray, q_valid: np.ndarray, k_cum: np.ndarray, k_valid: np.ndarray, device):
        """block mask (pi0_arch.py:22-28) as the kernels take it: q_limit[b,i] = #keys with cumsum <= the query's
        (keys are ordered, cumsum non-decreasing), key_valid[b,j] = input_mask.  Invalid queries get limit 0."""
        lim = (k_cum[:, None, :] <= q_cum[:, :, None]).sum(-1).astype(np.int32)
        lim[~q_valid] = 0
        return Pi0ForCausalLM._to_dev(lim, device), Pi0ForCausalLM._to_dev(k_valid.astype(np.uint8), device)

    @torch.no_grad()
    def _mot_forward(self, xs: List[Optional[torch.Tensor]], positions: Optional[np.ndarray], q_limit: torch.Tensor,
                     key_valid: torch.Tensor, past: Optional[list] = None, collect: bool = False, pos_parts=None, rope=None,
                     past_len: Optional[int] = None):
        """_inner_forward_mot (pi0_arch.py:116-216) without autograd: xs = [llm tokens | None, expert tokens | None],
        positions [B, S_q] int (RoPE), masks over [past keys ; new keys].  Returns ([out per expert], K/V cache)."""
        experts = [self.model.llm, self.model.action_expert]
        c = self.config.llm_config
        Hq, Hkv, D = c.num_attention_heads, c.num_key_value_heads, c.head_dim
        live = [(e, x) for e, x in zip(experts, xs) if x is not None]
        B = live[0][1].shape[0]
        lens = [x.shape[1] for _, x in live]
        S = sum(lens)
        dev = self.store.device
        offs = np.cumsum([0] + lens)
        if pos_parts is None:              # (the sampler hands in device tensors: nothing host-side inside its captured loop)
            rope = experts[0].rope_tables(int(positions.max()) + 1, dev)
            pos_parts = [self._to_dev(positions[:, offs[i]:offs[i + 1]].astype(np.int32), dev).reshape(-1) for i in range(len(live))]
        cos_t, sin_t = rope
        hs = [x.reshape(B * n, -1).contiguous() for (_, x), n in zip(live, lens)]
        cache = []
        for li in range(c.num_hidden_layers):
            qs, ks, vs = [], [], []
            # one request, one key/value head, one live expert, a cache with room behind the prefix (the sampler's Euler steps):
            # the new keys / values are written straight behind the cached ones — no concatenation (36 launches a step)
            kf, vf = past[li] if past is not None else (None, None)
            inplace = (past is not None and past_len is not None and len(live) == 1 and B == 1 and Hkv == 1
                       and kf.shape[2] == past_len + lens[0])
            for (e, _), h, n, pp in zip(live, hs, lens, pos_parts):
                if inplace:
                    q, k, v = K.rope_split(e.pre_attention(h, li), cos_t, sin_t, pp, B, n, Hq, Hkv, D,
                                           k_out=kf[:, :, past_len:], v_out=vf[:, :, past_len:])
                else:
                    q, k, v = K.rope_split(e.pre_attention(h, li), cos_t, sin_t, pp, B, n, Hq, Hkv, D)
                qs.append(q); ks.append(k); vs.append(v)
            q = qs[0] if len(qs) == 1 else torch.cat(qs, dim=2)
            k = ks[0] if len(ks) == 1 else torch.cat(ks, dim=2)
            v = vs[0] if len(vs) == 1 else torch.cat(vs, dim=2)
            if collect:
                cache.append((k, v))
            if inplace:
                k, v = kf, vf
            elif past is not None:
                k = torch.cat([kf[:, :, :past_len], k], dim=2)
                v = torch.cat([vf[:, :, :past_len], v], dim=2)
            o = torch.empty((B, S, Hq, D), device=dev, dtype=q.dtype)
            K.attn_fwd(q, k, v, o.permute(0, 2, 1, 3), causal=False, scale=D ** -0.5, q_limit=q_limit, key_valid=key_valid)
            nxt = []
            for i, ((e, _), h, n) in enumerate(zip(live, hs, lens)):
                a2 = o[:, offs[i]:offs[i + 1]].reshape(B * n, Hq * D)
                nxt.append(e.post_attention(h, a2.contiguous(), li))
            hs = nxt
        outs, it = [], iter(zip(live, hs, lens))
        for x in xs:
            if x is None:
                outs.append(None)
            else:
                (e, _), h, n = next(it)
                outs.append(e.final_norm(h).view(B, n, -1))
        return outs, cache

    def _mot_train(self, ptok, stok, positions, q_limit, key_valid) -> torch.Tensor:
        """the mixture with autograd: one Pi0MotLayerFn per layer over both experts; only the action expert's final
        norm is evaluated (prefix_out is never read by the loss, pi0_arch.py:372-388)"""
        llm, exp, st = self.model.llm, self.model.action_expert, self.store
        c = self.config.llm_config
        B, P, Sx = ptok.shape[0], ptok.shape[1], stok.shape[1]
        geom = (B, P, Sx, c.num_attention_heads, c.num_key_value_heads, c.head_dim)
        dev = st.device
        cos_t, sin_t = llm.rope_tables(int(positions.max()) + 1, dev)
        pos0 = self._to_dev(positions[:, :P].astype(np.int32), dev).reshape(-1)
        pos1 = self._to_dev(positions[:, P:].astype(np.int32), dev).reshape(-1)
        x0 = ptok.reshape(B * P, -1).contiguous()
        x1 = stok.reshape(B * Sx, -1).contiguous()
        n = c.num_hidden_layers
        for li in range(n):
            sp0, sp1 = llm.layer_specs[li], exp.layer_specs[li]
            x0, x1 = Fn.Pi0MotLayerFn.apply(x0, x1, st.params[sp1.down], st, sp0, sp1, geom, cos_t, sin_t, pos0, pos1,
                                            q_limit, key_valid, li == n - 1)
        y = Fn.NormFn.apply(x1, st.params[exp.p + "norm.weight"], st, "rms1p", exp.p + "norm.weight", None,
                            exp.config.rms_norm_eps)
        return y.view(B, Sx, -1)

    # ------------------------------------------------------------------------------------- training
    def forward(self, input_ids=None, attention_mask=None, actions=None, states=None, images=None, image_masks=None,
                **kwargs) -> CausalLMOutputDexbotic:
        """flow-matching step (pi0_arch.py:317-400).  kwargs ``noise`` [B,chunk,A] and ``time`` [B] inject the draws
        (reference: N(0,1) and Beta(1.5,1)*0.999+0.001)."""
        c, dev = self.config, self.store.device
        B = actions.shape[0]
        acts = actions.to(dev).float().reshape(B, c.chunk_size, c.action_dim)
        noise = kwargs.get("noise")
        noise = torch.randn_like(acts) if noise is None else noise.to(dev).float()
        time = kwargs.get("time")
        time = (np.random.beta(1.5, 1.0, size=B) * 0.999 + 0.001).astype(np.float32) if time is None else \
            np.asarray(time.cpu() if torch.is_tensor(time) else time, dtype=np.float32)
        # every mask and position of the mixture first: they depend on the two input masks only (the suffix is all valid), the
        # device masks are fetched while the stream still holds the previous step's tail, and the numpy below runs under it
        pmask = self.prefix_mask(attention_mask, image_masks)
        smask = np.ones((B, 1 + c.chunk_size), dtype=bool)
        sar = np.array([True, True] + [False] * (c.chunk_size - 1))
        input_mask = np.concatenate([pmask, smask], axis=1)
        cum = np.broadcast_to(np.cumsum(np.concatenate([np.zeros(pmask.shape[1], dtype=bool), sar]).astype(np.int64)), input_mask.shape)
        q_limit, key_valid = self._mask_tensors(cum, input_mask, cum, input_mask, dev)
        positions = np.cumsum(input_mask, axis=1) - 1
        te = self._to_dev(time, dev)[:, None, None]
        x_t = te * noise + (1 - te) * acts
        u_t = noise - acts
        ptok, pmask2, par = self.embed_prefix(input_ids, attention_mask, images, image_masks, input_mask=pmask)
        stok, smask2, sar2 = self.embed_suffix(states.to(dev).float(), x_t, time)
        assert smask2.shape == smask.shape and np.array_equal(sar2, sar) and not par.any()
        st = self.store
        if torch.is_grad_enabled():
            suf = self._mot_train(ptok, stok, positions, q_limit, key_valid)
        else:
            (_, suf), _ = self._mot_forward([ptok, stok], positions, q_limit, key_valid)
        v_t = Fn.LinearFn.apply(suf[:, -c.chunk_size:].reshape(B * c.chunk_size, -1).contiguous(),
                                st.params["model.action_out_proj.weight"], st, "model.action_out_proj.weight",
                                "model.action_out_proj.bias", L.ACT_NONE, None).view(B, c.chunk_size, -1).float()
        loss = Fn.MseLossFn.apply(v_t.contiguous(), u_t.contiguous())
        return CausalLMOutputDexbotic(loss=loss, logits=v_t)

    # ------------------------------------------------------------------------------------ inference
    @torch.no_grad()
    def inference_action(self, input_ids=None, attention_mask=None, states=None, images=None, image_masks=None,
                         diffusion_steps: int = 10, **kwargs):
        """pi0_arch.py:402-491.  kwarg ``noise`` [B,chunk,A] injects the initial sample.  Returns the [B,chunk,A]
        tensor (the exp layer de-normalises and slices, pi0_exp.py:484-514)."""
        c, dev, st = self.config, self.store.device, self.store
        B = states.shape[0]
        dt = -1.0 / diffusion_steps
        noise = kwargs.get("noise")
        x = (torch.randn(B, c.chunk_size, c.action_dim, device=dev) if noise is None else noise.to(dev)).float().contiguous()
        # host side first (masks, positions, the schedule's time embeddings: they depend on the input masks only), uploaded through
        # pinned memory — then the device work of the request is enqueued without a single wait on the stream
        pmask = self.prefix_mask(attention_mask, image_masks)
        pcum = np.zeros(pmask.shape, dtype=np.int64)
        p_limit, p_valid = self._mask_tensors(pcum, pmask, pcum, pmask, dev)
        ppos = np.cumsum(pmask, axis=1) - 1
        # ---- the Euler loop: everything that does not depend on x is prepared once (masks, positions, RoPE tables, the time
        #      embeddings of the whole schedule), the loop itself is tensors in / tensors out and is replayed as ONE HIP graph
        #      (graphs.GraphCache): 10 steps x 18 layers x ~14 tiny launches are host-bound when issued from Python
        smask = np.ones((B, 1 + c.chunk_size), dtype=bool)
        sar = np.array([True, True] + [False] * (c.chunk_size - 1))
        scum = np.broadcast_to(np.cumsum(sar.astype(np.int64)), smask.shape)
        # keys = [cached prefix (all visible where valid: cumsum 0) ; suffix]; queries = suffix (cumsum >= 1)
        k_cum = np.concatenate([np.zeros_like(pmask, dtype=np.int64), scum], axis=1)
        k_valid = np.concatenate([pmask, smask], axis=1)
        q_limit, key_valid = self._mask_tensors(scum, smask, k_cum, k_valid, dev)
        fpos = pmask.sum(-1)[:, None] + np.cumsum(smask, axis=-1) - 1
        rope = self.model.llm.rope_tables(int(fpos.max()) + 1, dev)
        pos = self._to_dev(fpos.astype(np.int32), dev).reshape(-1)
        times, time = [], np.float32(1.0)
        while time > -dt / 2:                                             # the reference's float32 schedule (pi0_arch.py:470-489)
            times.append(time)
            time = np.float32(time + np.float32(dt))
        da = c.action_config.hidden_size
        te_table = self._to_dev(np.stack([posemb_sincos(np.full(B, t, dtype=np.float32), da) for t in times]), dev
                                ).to(st.compute_dtype)                                            # [steps, B, da]
        ptok, _, par = self.embed_prefix(input_ids, attention_mask, images, image_masks, input_mask=pmask)
        assert not par.any()
        _, cache = self._mot_forward([ptok, None], ppos, p_limit, p_valid, collect=True)
        states_d = states.to(dev).float().contiguous()
        n_layers = len(cache)

        P_len, Sx = int(cache[0][0].shape[2]), int(smask.shape[1])

        def euler(x, states_d, te_table, q_limit, key_valid, pos, **kv):
            # per layer ONE buffer [prefix keys ; room for the suffix keys], filled behind the prefix by every step's RoPE kernel
            past = [tuple(torch.cat([kv[f"{n}{i}"], kv[f"{n}{i}"].new_zeros(B, kv[f"{n}{i}"].shape[1], Sx, kv[f"{n}{i}"].shape[3])], dim=2)
                          for n in ("k", "v")) for i in range(n_layers)]
            for s in range(len(times)):
                stok, _, _ = self.embed_suffix(states_d, x, None, te=te_table[s])
                (_, suf), _ = self._mot_forward([None, stok], None, q_limit, key_valid, past=past, pos_parts=[pos], rope=rope,
                                                past_len=P_len)
                v_t = Fn.LinearFn.apply(suf[:, -c.chunk_size:].reshape(B * c.chunk_size, -1).contiguous(),
                                        st.params["model.action_out_proj.weight"], st, "model.action_out_proj.weight",
                                        "model.action_out_proj.bias", L.ACT_NONE, None).view(B, c.chunk_size, -1).float()
                x = K.add(x, K.scale_(v_t.contiguous(), dt))             # Euler step x += v dt
            return x
        inputs = dict(x=x, states_d=states_d, te_table=te_table, q_limit=q_limit, key_valid=key_valid, pos=pos)
        for i, (k_, v_) in enumerate(cache):
            inputs[f"k{i}"], inputs[f"v{i}"] = k_.contiguous(), v_.contiguous()
        from ... import graphs
        if dev.type == "cuda" and kwargs.get("use_graph", graphs.enabled()):
            gc_ = self.__dict__.setdefault("_sampler_graphs", graphs.GraphCache(dev))
            return gc_.run(("euler", int(diffusion_steps), int(fpos.max()), rope[0].data_ptr()), euler, inputs).clone()
        return euler(**inputs)


from ..dexbotic_arch import register_model_with_hf  # noqa: E402

register_model_with_hf(Pi0ForCausalLM)
