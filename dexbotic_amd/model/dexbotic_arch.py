"""Base VLM classes: host-side mirror of dexbotic/model/dexbotic_arch.py on libdexbotic_amd kernels.

Same public surface as the reference: ``DexboticConfig`` (:17-23),
``CausalLMOutputDexbotic`` (:26-34), ``DexboticVLMModel`` with the ``_build_*`` factories, module /
prefix properties, ``_extract_vision_features`` (:157-180) and
``_prepare_inputs_labels_for_multimodal`` (:182-373), ``DexboticForCausalLM`` (:415-542) and
``ActionOutputForCausalLM._denorm`` (:546-563).  What differs is underneath: parameters live in the
flat arenas of engine.ParamStore, arithmetic is libdexbotic_amd.so.
"""
from __future__ import annotations

import json
import os
from abc import ABC, abstractmethod
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from .. import functional as Fn
from .. import hostcpu
from .. import kernels as K
from ..constants import IGNORE_INDEX
from ..engine import ParamStore, attach_parameters, building
from ..splice import PlanCache, SplicePlan, build_splice_plan
from .llm.qwen2 import Qwen2Backbone, Qwen2Config
from .modules.mm_projector.builder import build_vision_projector
from .modules.mm_vision.builder import build_vision_tower

_DTYPES = {"float32": torch.float32, "bfloat16": torch.bfloat16, torch.float32: torch.float32,
           torch.bfloat16: torch.bfloat16}


def _hf_config_base():
    from transformers import PretrainedConfig
    return PretrainedConfig


class DexboticConfig(_hf_config_base()):
    """``transformers.PretrainedConfig`` subclass like the reference's (dexbotic_arch.py:17-23), registered with
    ``AutoConfig`` under the same ``model_type`` string, so ``AutoConfig.from_pretrained(ckpt)`` / ``config.json`` written
    by either implementation resolve to it.  ``llm_config`` may be a Qwen2Config, a dict, an HF config object or a
    directory with config.json; its keys are merged in (``_merge_llm``, dexbotic_arch.py:79-85) so ``hidden_size`` /
    ``vocab_size`` are top-level."""
    model_type = "dexbotic"

    def __init__(self, llm_config=None, mm_projector_type: Optional[str] = "mlp2x_gelu", mm_vision_tower=None,
                 chat_template: Optional[str] = "dexbotic", init_llm_weights: bool = False,
                 compute_dtype="float32", **kwargs):
        if isinstance(llm_config, str):
            with open(os.path.join(llm_config, "config.json")) as f:
                llm_config = json.load(f)
        self.llm_config = Qwen2Config.from_any(llm_config if llm_config is not None else {})
        self.mm_projector_type = mm_projector_type
        self.mm_vision_tower = mm_vision_tower
        self.chat_template = chat_template
        self.init_llm_weights = False
        self.compute_dtype = compute_dtype if isinstance(compute_dtype, str) else str(compute_dtype).replace("torch.", "")
        self.tokenizer_model_max_length = kwargs.pop("tokenizer_model_max_length", None)
        self.tokenizer_padding_side = kwargs.pop("tokenizer_padding_side", "right")
        self.image_aspect_ratio = kwargs.pop("image_aspect_ratio", "pad")
        # fp32 x fp32 products of the fp32 action head: "bf16x3" (split-bf16 on the MFMA ring kernel, the counterpart of
        # the reference's tf32=True, base_exp.py:254) in bf16 compute mode, exact fp32 MFMA in fp32 (parity) mode
        self.fp32_matmul = kwargs.pop("fp32_matmul", None) or ("bf16x3" if "bfloat16" in self.compute_dtype else "exact")
        kwargs.pop("model_type", None)
        merged = {k: kwargs.pop(k) for k in list(kwargs) if k in self.llm_config.to_dict() and k != "model_type"}
        super().__init__(**kwargs)
        for k, v in self.llm_config.to_dict().items():          # _merge_llm: only add missing keys
            if k in ("model_type", "architectures", "transformers_version", "torch_dtype", "dtype"):
                continue
            if k in merged:
                setattr(self, k, merged[k])
            elif k not in self.__dict__:
                setattr(self, k, v)

    # ---- (de)serialisation compatible with the reference's config.json ------------------------------
    def to_dict(self) -> Dict[str, Any]:
        d = super().to_dict()
        out = {}
        for k, v in d.items():
            if hasattr(v, "to_dict"):
                v = v.to_dict()
            elif isinstance(v, torch.dtype):
                v = str(v).replace("torch.", "")
            out[k] = v
        out["model_type"] = self.model_type
        return out

    def to_diff_dict(self) -> Dict[str, Any]:
        return self.to_dict()

    @classmethod
    def from_pretrained(cls, path: str, **kwargs):
        with open(os.path.join(path, "config.json")) as f:
            return cls.from_dict(json.load(f), **kwargs)

    @classmethod
    def from_dict(cls, d: Dict[str, Any], **kwargs):
        """HF contract: with ``return_unused_kwargs=True`` (AutoModel.from_pretrained asks for it) -> (config, unused kwargs)"""
        d = dict(d)
        for k in ("model_type", "architectures", "transformers_version"):
            d.pop(k, None)
        config = cls(**d)
        if kwargs.pop("return_unused_kwargs", False):
            unused = {k: v for k, v in kwargs.items() if not k.startswith("_") and k not in ("name_or_path", "trust_remote_code")}
            return config, unused
        return config

    def save_pretrained(self, path: str, **kwargs) -> None:
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, "config.json"), "w") as f:
            json.dump(self.to_dict(), f, indent=2, default=str)


def register_with_hf(config_cls) -> None:
    """AutoConfig.register under the reference's model_type strings (dexbotic_arch.py:18, cogact_arch.py:14, ...)"""
    from transformers import AutoConfig
    try:
        AutoConfig.register(config_cls.model_type, config_cls)
    except ValueError:
        pass                                         # already registered (module re-import)


def register_model_with_hf(model_cls) -> None:
    """AutoModel.register(Config, ForCausalLM), the pattern of dexbotic/model/dm0/__init__.py:12-16 and pi05/__init__.py:6-7:
    ``AutoModel.from_pretrained(ckpt)`` on a reference checkpoint directory builds the native class"""
    from transformers import AutoModel
    try:
        AutoModel.register(model_cls.config_class, model_cls)
    except ValueError:
        pass                                         # already registered (module re-import)


register_with_hf(DexboticConfig)


@dataclass
class CausalLMOutputDexbotic:
    loss: Optional[torch.Tensor] = None
    logits: Optional[torch.Tensor] = None
    past_key_values: Optional[Any] = None
    hidden_states: Optional[Tuple[torch.Tensor, ...]] = None
    attentions: Optional[Tuple[torch.Tensor, ...]] = None
    text_loss: Optional[torch.Tensor] = None
    action_loss: Optional[torch.Tensor] = None

    def __getitem__(self, k):
        return getattr(self, k) if isinstance(k, str) else tuple(v for v in self.__dict__.values() if v is not None)[k]

    def keys(self):
        return [k for k, v in self.__dict__.items() if v is not None]


class DexboticVLMModel(nn.Module):
    """vision tower -> projector -> splice into LLM embeddings -> LLM backbone."""
    # dexbotic_arch.py:40.  Off by default (25 GB of kept activations out of 288 at the CogACT batch);
    # gradient_checkpointing_enable() on the causal-LM wrapper turns on ParamStore.recompute (functional.py)
    supports_gradient_checkpointing = True

    def __init__(self, config: DexboticConfig, store: ParamStore):
        super().__init__()
        self.config = config
        self.store = store
        # registration order = forward order = arena order (the DP reducer walks it backwards)
        self.mm_vision_tower = None
        self.mm_projector = None
        if getattr(config, "mm_vision_tower", None) is not None:
            self.mm_vision_tower = self._build_mm_vision_module(config.mm_vision_tower)
            self.mm_projector = self._build_mm_projector_module(config)
        self.llm = Qwen2Backbone(store, "model.llm.", config.llm_config)
        self._last_plan: Optional[SplicePlan] = None
        self._plans = PlanCache()

    def initialize_model(self, extra_config: dict):
        for key, value in extra_config.items():
            setattr(self.config, key, value)

    def _build_mm_projector_module(self, config) -> nn.Module:
        if getattr(self, "mm_projector", None) is not None:
            return self.mm_projector
        with building(self.store):
            self.mm_projector = build_vision_projector(config)
        return self.mm_projector

    def _build_mm_vision_module(self, config) -> nn.Module:
        if getattr(self, "mm_vision_tower", None) is not None:
            return self.mm_vision_tower
        with building(self.store):
            self.mm_vision_tower = build_vision_tower(config)
        self.config.mm_hidden_size = self.mm_vision_tower.hidden_size
        return self.mm_vision_tower

    @property
    def mm_projector_module(self) -> nn.Module:
        return self.mm_projector

    @property
    def mm_projector_prefix(self) -> str:
        return "mm_projector"

    @property
    def mm_vision_module(self) -> nn.Module:
        return self.mm_vision_tower

    @property
    def mm_vision_prefix(self) -> str:
        return "mm_vision"

    @property
    def backbone(self) -> nn.Module:
        return self.llm

    @property
    def device(self):
        return self.store.device

    @property
    def dtype(self):
        return self.store.compute_dtype

    def _extract_vision_features(self, images: torch.Tensor) -> torch.Tensor:
        """[B,3,H,W] or [B,V,3,H,W] -> [B, V*N_v, d]; the views of a sample are concatenated along tokens."""
        if images.ndim == 5:
            B, V = images.shape[:2]
            feats = self.mm_projector_module(self.mm_vision_module(images.flatten(0, 1)))
            return feats.view(B, V * feats.shape[1], feats.shape[2])
        return self.mm_projector_module(self.mm_vision_module(images))

    def num_image_tokens(self, images: torch.Tensor) -> int:
        """tokens one sample's image placeholder expands to (views concatenated along tokens)"""
        views = images.shape[1] if images.ndim == 5 else 1
        return views * self.mm_vision_module.num_patches

    def _prepare_inputs_labels_for_multimodal(self, input_ids, position_ids, attention_mask, past_key_values, labels,
                                              cache_position, images) -> tuple:
        """Same contract as the reference (returns input_ids=None and the spliced inputs_embeds); the plan
        (kv ranges, last-token index) is kept in ``self._last_plan`` for the backbone / cognition gather."""
        if self.mm_vision_module is None or images is None:
            raise NotImplementedError("text-only forward is outside the VLA path")
        # the integer plan first (host arithmetic on a few KB; ids handed over as host tensors — the collator's own
        # output — or a batch object seen before cost no device sync), THEN the vision tower is enqueued
        plan = self._plans.get(input_ids, attention_mask, labels, self.num_image_tokens(images),
                               getattr(self.config, "tokenizer_model_max_length", None),
                               getattr(self.config, "tokenizer_padding_side", "right"))
        self._last_plan = plan
        image_features = self._extract_vision_features(images)                          # [B, V*N_v, d]
        dev = image_features.device
        pd = plan.dev(dev)
        B, S = plan.plan.shape
        embeds = Fn.SpliceFn.apply(image_features, self.store.params[self.llm.embed_name], self.store,
                                   self.llm.embed_name, pd["plan"]).view(B, S, -1)
        # copies (a few KB): callers may edit labels / mask in place (HF-style loss code does); the plan cache keeps its own
        new_labels = None if labels is None else pd["labels"].clone()
        new_mask = None
        if attention_mask is not None:
            new_mask = pd["mask"].clone() if attention_mask.dtype == torch.bool else pd["mask"].to(attention_mask.dtype)
        return None, position_ids, new_mask, past_key_values, embeds, new_labels, cache_position

    def run_llm(self, inputs_embeds: torch.Tensor, attention_mask: Optional[torch.Tensor]) -> torch.Tensor:
        """HF Qwen2Model(inputs_embeds, attention_mask) -> hidden_states[-1]; key-padding ranges come from the
        splice plan (causal AND padding mask, position_ids = arange(S))."""
        kv_start = kv_end = None
        plan = self._last_plan
        if attention_mask is not None and plan is not None and not plan.attention_mask.all():
            pd = plan.dev(inputs_embeds.device)
            kv_start, kv_end = pd["kv_start"], pd["kv_end"]
        return self.llm(inputs_embeds, kv_start, kv_end)


class NativePreTrainedMixin:
    """from_pretrained / save_pretrained / state_dict plumbing shared by the *ForCausalLM classes."""
    config_class = DexboticConfig

    def _finish_init(self, train: bool) -> None:
        self.store.finalize(train=train)
        attach_parameters(self, self.store)
        self.register_forward_pre_hook(NativePreTrainedMixin._external_loop_prelude)
        self.register_forward_hook(NativePreTrainedMixin._external_loop_epilogue)
        self.register_state_dict_pre_hook(lambda *a, **k: self.store.wait_pending())

    @staticmethod
    def _external_loop_prelude(self, args) -> None:
        """training forward of a loop this package does not manage (HF Trainer.training_step, the reference's
        DexboticTrainer, a hand-written loop): ParamStore.external_prelude does what NativeTrainer would have done"""
        st = self.store
        if st.managed or st.grad is None or not self.training or not torch.is_grad_enabled():
            return
        unused = self.unused_parameter_names() if hasattr(self, "unused_parameter_names") else ()
        st.external_prelude(unused)
        # the fp32 head products run as the model's config says (bf16x3 under bf16 compute; NativeTrainer.micro_step scopes its
        # forward and backward the same way).  Scoped to this forward (_external_loop_epilogue); the backward of every Function
        # re-enters the mode its forward ran in (functional._StoreFn)
        from .. import kernels as K
        stale = self.__dict__.pop("_f32_mode_prev", None)
        if stale is not None:
            K.F32_GEMM_MODE = stale          # a previous forward raised before its epilogue hook ran: undo its mode first
        self.__dict__["_f32_mode_prev"] = K.F32_GEMM_MODE
        K.F32_GEMM_MODE = getattr(self.config, "fp32_matmul", "exact")

    @staticmethod
    def _external_loop_epilogue(self, args, output) -> None:
        prev = self.__dict__.pop("_f32_mode_prev", None)
        if prev is not None:
            from .. import kernels as K
            K.F32_GEMM_MODE = prev

    def train(self, mode: bool = True):
        out = nn.Module.train(self, mode)
        if not mode:
            self.store.external_eval()
        else:
            # captured inference graphs hold pointers to eval-time caches (e.g. Gemma's 1 + w norm weights): drop them
            self.__dict__.pop("_sampler_graphs", None)
        return out

    def _apply(self, fn, recurse: bool = True):
        """``model.to(device)`` / ``.cuda()`` (HF Trainer moves the model in its constructor) must not rebuild the parameters:
        they are views of the arenas.  A conversion that would change nothing is accepted and ignored, anything else refused."""
        probe = fn(torch.empty(0, device=self.store.device, dtype=torch.float32))
        if probe.device.type != self.store.device.type or probe.dtype != torch.float32:
            raise NotImplementedError("dexbotic_amd: the parameters live in device arenas built at construction "
                                      f"({self.store.device}, fp32 masters + {self.store.compute_dtype} shadows); "
                                      "construct the model with device= / compute_dtype instead of converting it")
        return self

    def zero_grad(self, set_to_none: bool = True) -> None:
        """HF Trainer calls ``model.zero_grad()`` after every optimizer step.  The gradients are views of the arena and the
        next backward's first write replaces them, so nothing is zeroed or dropped: only the step boundary is recorded."""
        if self.store.grad is None:
            return
        self.store.external_zero_grad()
        if self.store.managed:
            return
        if not set_to_none:
            self.store.grad.zero_()

    def post_load(self) -> None:
        self.store.sync_shadow()

    supports_gradient_checkpointing = True      # dexbotic_arch.py:40

    def gradient_checkpointing_enable(self, gradient_checkpointing_kwargs=None) -> None:
        """HF ``PreTrainedModel.gradient_checkpointing_enable`` as base_exp.py:245 / trainer.py:101,120 switch it on (HF then
        checkpoints every Qwen2 decoder layer and every CLIP encoder layer: only the layer input is kept and the layer's forward
        runs again inside its backward).  Native form: ``ParamStore.recompute`` — the decoder / vision / pi0 layer Functions keep
        their INPUT only and re-run their forward launches at the top of their backward (functional.Qwen2LayerFn / VitBlockFn /
        Pi0MotLayerFn; same kernels in the same order, so the gradients are bit-identical to the resident-activation step).
        Kept activations drop from ~0.75 GB to 33 MB per decoder layer at 16 x 287 tokens, for one more forward per step.
        ``gradient_checkpointing_kwargs`` (``use_reentrant``) has no meaning here and is accepted.  Resident activations remain
        the default: on a 288 GB part the CogACT recipe keeps 25 GB of them; DEXBOTIC_AMD_ACCEPT_GRAD_CHECKPOINTING=1 /
        ``config.accept_gradient_checkpointing`` keep their old meaning (accept the call, stay resident) for exp scripts whose
        TrainerConfig default asks for checkpointing only to fit 80 GB parts."""
        if os.environ.get("DEXBOTIC_AMD_ACCEPT_GRAD_CHECKPOINTING", "0") != "0" or \
                getattr(self.config, "accept_gradient_checkpointing", False):
            import warnings
            warnings.warn("dexbotic_amd: gradient_checkpointing requested and ignored (activations stay resident in HBM)")
            return None
        self.store.recompute = True

    def gradient_checkpointing_disable(self) -> None:
        self.store.recompute = False

    @property
    def is_gradient_checkpointing(self) -> bool:
        return bool(self.store.recompute)

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        # checkpoints written under transformers 4.51 carry ".vision_tower.vision_model." 
        sd = {k.replace(".vision_tower.vision_model.", ".vision_tower."): v for k, v in state_dict.items()}
        out = nn.Module.load_state_dict(self, sd, strict=strict)
        self.post_load()
        return out

    @classmethod
    def from_pretrained(cls, path: str, torch_dtype=None, device=None, train: bool = False, **kw):
        from safetensors.torch import load_file
        config = cls.config_class.from_pretrained(path)
        if torch_dtype is not None:
            config.compute_dtype = str(torch_dtype).replace("torch.", "")
        model = cls(config, device=device, train=train)
        files = sorted(f for f in os.listdir(path) if f.endswith(".safetensors"))
        sd = {}
        for f in files:
            sd.update(load_file(os.path.join(path, f)))
        model.load_state_dict(sd, strict=True)
        return model

    def save_pretrained(self, path: str) -> None:
        from safetensors.torch import save_file
        self.config.save_pretrained(path)
        save_file({k: v.detach().cpu().contiguous() for k, v in self.state_dict().items()},
                  os.path.join(path, "model.safetensors"))

    @torch.no_grad()
    def init_random_(self, seed: int = 0, std: float = 0.02) -> None:
        """synthetic weights for benchmarks (no checkpoints offline): N(0, std) matrices, unit norm weights,
        zero biases — same families HF's _init_weights uses for these modules."""
        g = torch.Generator(device=self.store.device)
        g.manual_seed(seed)
        self.store.master.normal_(0.0, std, generator=g)
        for name in self.store.slots:
            t = self.store.w32(name)
            leaf = name.rsplit(".", 1)[-1]
            if leaf == "bias":
                t.zero_()
            elif leaf == "weight" and t.dim() == 1:
                t.fill_(1.0)
        self.post_load()


class DexboticForCausalLM(NativePreTrainedMixin, nn.Module):
    config_class = DexboticConfig

    def __init__(self, config: DexboticConfig, device=None, train: bool = True):
        super().__init__()
        self.config = config
        config.model_type = self.config_class.model_type
        device = device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu")
        self.store = ParamStore(device, _DTYPES[config.compute_dtype])
        with building(self.store):                    # the one-argument factories register into this arena
            self._real_init(config)
        self._finish_init(train)

    def _real_init(self, config):
        self.model = DexboticVLMModel(config, self.store)
        self.store.new_bucket()
        self.store.register([("lm_head.weight", (config.vocab_size, config.hidden_size))])

    @property
    def device(self):
        return self.store.device

    @property
    def dtype(self):
        return self.store.compute_dtype

    # DexboticTrainer.create_optimizer hands the *ForCausalLM to OptimizerConfig._get_optimizer_grouped_parameters, which reads
    # the prefix properties off it when a module has a learning rate of its own (trainer.py:25-36, base_exp.py:111-160)
    @property
    def mm_projector_prefix(self) -> str:
        return self.model.mm_projector_prefix

    @property
    def mm_vision_prefix(self) -> str:
        return self.model.mm_vision_prefix

    @property
    def action_head_prefix(self) -> str:
        return self.model.action_head_prefix

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, inputs_embeds=None,
                labels=None, use_cache=None, output_attentions=None, output_hidden_states=None, images=None,
                return_dict=None, cache_position=None, actions=None, states=None) -> CausalLMOutputDexbotic:
        (_, position_ids, attention_mask, past_key_values, inputs_embeds, labels, cache_position
         ) = self.model._prepare_inputs_labels_for_multimodal(input_ids, position_ids, attention_mask, past_key_values,
                                                               labels, cache_position, images)
        hidden = self.model.run_llm(inputs_embeds, attention_mask)
        B, S, d = hidden.shape
        if labels is None:
            with torch.no_grad():
                logits = K.mm_nt(hidden.reshape(B * S, d).contiguous(), self.store.w("lm_head.weight")).view(B, S, -1)
            return CausalLMOutputDexbotic(loss=None, logits=logits, hidden_states=(hidden,))
        # HF ForCausalLMLoss (dexbotic_arch.py:488): position t is scored against label t+1, the last one is ignored
        lab = self.model._last_plan.labels
        shifted = np.full_like(lab, IGNORE_INDEX)
        shifted[:, :-1] = lab[:, 1:]
        n_valid = int((shifted != IGNORE_INDEX).sum())
        loss, logits = Fn.LmHeadLossFn.apply(hidden, self.store.params["lm_head.weight"], self.store, "lm_head.weight",
                                             hostcpu.upload(shifted.reshape(-1), hidden.device), n_valid)
        return CausalLMOutputDexbotic(loss=loss, logits=logits, hidden_states=(hidden,))

    def unused_parameter_names(self) -> List[str]:
        """parameters that get no gradient from the LM loss (the CLIP layer after hidden_states[-2], post_layernorm)"""
        return self.model.mm_vision_tower.unused_parameter_names()

    @torch.no_grad()
    def generate(self, input_ids, images=None, max_new_tokens: int = 64, do_sample: bool = False,
                 temperature: float = 1.0, eos_token_id: Optional[int] = None, stopping_criteria=None,
                 return_dict_in_generate: bool = False, generator: Optional[torch.Generator] = None,
                 attention_mask=None, **kwargs):
        """Token-by-token continuation over a KV cache — the subset of GenerationMixin.generate the reference uses
        (discrete_vla_arch.py:33-41: batch 1, greedy or temperature sampling, stopping criteria on the decoded tail).
        Prefill = vision tower + splice + decoder with the cache filled; each step = one cached decoder pass on the
        new token, lm_head on its hidden state, argmax (first maximal index) or a multinomial draw."""
        dev = self.store.device
        imgs = images.to(device=dev, dtype=self.store.compute_dtype)
        feats = self.model._extract_vision_features(imgs)
        plan = build_splice_plan(input_ids.detach().cpu().numpy(),
                                 None if attention_mask is None else attention_mask.detach().cpu().numpy().astype(bool),
                                 None, feats.shape[1],
                                 getattr(self.config, "tokenizer_model_max_length", None),
                                 getattr(self.config, "tokenizer_padding_side", "right"))
        B, S = plan.plan.shape
        pad = None
        if not plan.attention_mask.all():
            # prompts of unequal length: HF's generate wants them LEFT padded (new tokens continue every prompt directly);
            # every sample then attends to its own key range [pad_b, ...) of the cache and counts positions from its
            # first real token.  Right-padded batches leave a hole between prompt and continuation: refused.
            if not (plan.attention_mask[:, -1].all() and
                    all(plan.attention_mask[b, int(plan.kv_start[b]):].all() for b in range(B))):
                raise ValueError("generate(): a batch of unequal-length prompts must be LEFT padded "
                                 "(tokenizer.padding_side = 'left'), as with HF generate")
            pad = [int(v) for v in plan.kv_start]
        llm = self.model.llm
        embed = self.store.params[llm.embed_name]
        x = Fn.SpliceFn.apply(feats, embed, self.store, llm.embed_name,
                              torch.from_numpy(plan.plan.reshape(-1)).to(dev)).view(B, S, -1)
        W_lm, W_emb = self.store.w("lm_head.weight"), self.store.w(llm.embed_name)
        # (round 5: the decode step captured once into a HIP graph and replayed per token — rotary row, cache slot and key range in
        #  device tensors — measured SLOWER than this eager loop, 4.84 against 4.48 ms/token at full size: the step is bound by its
        #  ~340 short kernels on the GPU, not by the host that issues them; profiles/r05_decode_graph.txt.  Not kept.)
        cache = llm.new_cache(B, S + max_new_tokens, dev, x.dtype)
        last = llm.forward_cached(x, cache, pad)[:, -1].contiguous()
        seq = input_ids.to(dev)
        new_tokens, step_logits = [], []
        logits = K.mm_nt(last, W_lm)                                           # [B, V]
        for t in range(max_new_tokens):
            if do_sample:
                probs = torch.softmax(logits.float() / max(temperature, 1e-6), dim=-1)
                nxt = torch.multinomial(probs, 1, generator=generator).view(-1)
            else:
                nxt = K.argmax_rows(logits)
            new_tokens.append(nxt)
            if kwargs.get("output_logits"):
                step_logits.append(logits.float())
            seq = torch.cat([seq, nxt.view(B, 1)], dim=1)
            done = eos_token_id is not None and bool((nxt == eos_token_id).all())
            if not done and stopping_criteria:
                done = any(bool(torch.as_tensor(sc(seq, None)).all()) for sc in stopping_criteria)
            if done or t + 1 == max_new_tokens:
                break
            logits = K.mm_nt(llm.forward_cached(W_emb[nxt].view(B, 1, -1), cache, pad)[:, -1].contiguous(), W_lm)
        if cache.fused_steps and K.decode_timed_out():
            # (the host has synchronised on every token anyway: stopping criteria / eos read the ids)
            raise RuntimeError("generate(): a persistent decode launch gave up at a device-wide barrier (something else held CUs for "
                               "seconds); its tokens are garbage — re-run, or set DXA_DECODE_FUSED=0 for the per-op decode step")
        if return_dict_in_generate:
            return GenerateOutput(sequences=seq, logits=tuple(step_logits) if step_logits else None)
        return seq

    def process_images(self, images):
        """dexbotic_arch.py:498-529 on the device: the uint8 frames are uploaded as they are and libdexbotic_amd pads,
        resizes (Pillow-exact bicubic), crops and normalises them (data/dataset/rgb_preprocess.py).  Frames of equal
        size share one launch; the result is stacked when all outputs have one shape, as in the reference."""
        from ..data.dataset.rgb_preprocess import PreprocessRGB, to_uint8_hwc
        proc = self.model.mm_vision_module.image_processor
        aspect = getattr(self.config, "image_aspect_ratio", "pad")
        pre = PreprocessRGB(proc, image_aspect_ratio="pad" if aspect == "pad" else None, device=self.device)
        frames = [to_uint8_hwc(im) for im in images]
        if frames and all(f.shape == frames[0].shape for f in frames):
            if all(f.device.type == "cpu" for f in frames):
                # host frames are stacked by numpy (one memcpy per frame): torch.stack above ATen's parallel grain wakes the whole
                # OpenMP pool for 393 KB, and its spinning workers cost a serving process its cgroup CPU quota (hostcpu.py)
                import numpy as np
                return pre.batch(torch.from_numpy(np.stack([f.numpy() for f in frames])))
            return pre.batch(torch.stack(frames))
        out = [pre.batch(f[None])[0] for f in frames]
        if out and all(x.shape == out[0].shape for x in out):
            out = torch.stack(out, dim=0)
        return out

    @staticmethod
    def expand2square(pil_img, background_color):
        from PIL import Image
        w, h = pil_img.size
        if w == h:
            return pil_img
        side = max(w, h)
        canvas = Image.new(pil_img.mode, (side, side), background_color)
        canvas.paste(pil_img, ((side - w) // 2, (side - h) // 2))
        return canvas


register_model_with_hf(DexboticForCausalLM)


@dataclass
class GenerateOutput:
    sequences: torch.Tensor
    logits: Optional[tuple] = None


class ActionOutputForCausalLM(ABC):
    @abstractmethod
    def inference_action(self, input_ids, image_tensor, inference_args={}, **kwargs):
        ...

    def _denorm(self, actions, action_norms) -> np.ndarray:
        """[-1,1] -> physical units; host numpy like the reference (bit-exact contract)."""
        lo = np.array(action_norms["min"]).reshape(1, -1)
        hi = np.array(action_norms["max"]).reshape(1, -1)
        a = np.clip(actions, -1, 1)
        return lo + (a + 1) * 0.5 * (hi - lo)
