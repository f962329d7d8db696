"""DB-CogACT policy: host-side mirror of dexbotic/model/cogact/cogact_arch.py on libdexbotic_amd kernels.

``CogActConfig`` (:13-17), ``CogActModel`` (:20-45), ``CogACTForCausalLM.forward`` (:56-147: VLM prefill ->
cognition feature of the last un-padded token -> 4x repeated diffusion loss in fp32) and
``.inference_action`` (:149-198: prefill -> CFG 1.5 -> 10-step DDIM -> de-normalised [T][A] list).
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np
import torch
import torch.nn as nn

from ... import functional as Fn
from ..dexbotic_arch import (ActionOutputForCausalLM, CausalLMOutputDexbotic, DexboticConfig, DexboticForCausalLM,
                             DexboticVLMModel)
from .action_model.builder import build_action_model


class CogActConfig(DexboticConfig):
    model_type = "dexbotic_cogact"

    def __init__(self, action_model_type: Optional[str] = None, action_dim: Optional[int] = None,
                 chunk_size: Optional[int] = None, **kwargs):
        super().__init__(**kwargs)
        self.action_model_type = action_model_type
        self.action_dim = action_dim
        self.chunk_size = chunk_size


class CogActModel(DexboticVLMModel):
    def __init__(self, config: CogActConfig, store):
        super().__init__(config, store)
        self.action_head = None
        if config.action_model_type is not None:
            self.action_head = self._build_action_head_module(config)

    def _build_action_head_module(self, config: CogActConfig):
        if getattr(self, "action_head", None) is not None:
            return self.action_head
        self.action_head = build_action_model(config, self.store, "model.action_head.")
        return self.action_head

    @property
    def action_head_module(self) -> nn.Module:
        return self.action_head

    @property
    def action_head_prefix(self) -> str:
        return "action_head"


class CogACTForCausalLM(DexboticForCausalLM, ActionOutputForCausalLM):
    config_class = CogActConfig

    def _real_init(self, config: CogActConfig):
        self.model = CogActModel(config, self.store)
        self.store.new_bucket()
        self.store.register([("lm_head.weight", (config.vocab_size, config.hidden_size))])

    def unused_parameter_names(self) -> List[str]:
        """parameters that never get a gradient in CogACT training (lm_head, the CLIP layer after
        hidden_states[-2], post_layernorm, history_embedder): skipped by the DP reducer."""
        names = ["lm_head.weight"] + self.model.mm_vision_tower.unused_parameter_names()
        names += [n for n in self.store.slots if ".history_embedder." in n]
        return names

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, inputs_embeds=None,
                labels=None, use_cache=None, output_attentions=None, output_hidden_states=None, images=None,
                return_dict=None, cache_position=None, actions=None, states=None, repeated_diffusion_steps: int = 4,
                **kwargs) -> CausalLMOutputDexbotic:
        """kwargs ``noise`` [R*B,T,A], ``timesteps`` [R*B], ``drop_ids`` [R*B] inject the random draws of the
        action loss (parity tests); by default they are drawn like the reference does."""
        (_, position_ids, attention_mask, past_key_values, inputs_embeds, labels, cache_position
         ) = self.model._prepare_inputs_labels_for_multimodal(input_ids, position_ids, attention_mask, past_key_values,
                                                               labels, cache_position, images)
        last_hidden_state = self.model.run_llm(inputs_embeds, attention_mask)            # [B,S,d], post final norm
        B, S, d = last_hidden_state.shape
        loss = None
        if attention_mask is not None and actions is not None:
            plan = self.model._last_plan
            idx = torch.from_numpy(np.arange(B, dtype=np.int64) * S + plan.last_index).to(last_hidden_state.device)
            cognition = Fn.GatherRowsFn.apply(last_hidden_state.reshape(B * S, d), idx)      # [B,d] fp32
            A, T = self.config.action_dim, self.config.chunk_size
            acts = actions.reshape(actions.size(0), -1, A).float()[:, :T, :]
            R = repeated_diffusion_steps
            acts_rep = acts.repeat(R, 1, 1)
            cog_rep = cognition.repeat(R, 1).unsqueeze(1)                                   # [R*B,1,d]
            loss = self.model.action_head_module.loss(acts_rep, cog_rep, noise=kwargs.get("noise"),
                                                      timestep=kwargs.get("timesteps"), drop_ids=kwargs.get("drop_ids"))
        return CausalLMOutputDexbotic(loss=loss, logits=last_hidden_state, past_key_values=None,
                                      hidden_states=(last_hidden_state,), attentions=None)

    __call__ = nn.Module.__call__

    @torch.no_grad()
    def inference_action(self, input_ids, image_tensor, inference_args={}, **kwargs):
        cfg_scale = inference_args.get("cfg_scale", 1.5)
        num_ddim_steps = inference_args.get("num_ddim_steps", 10)
        action_norms = inference_args.get("action_norms")
        out = self(input_ids=input_ids, images=image_tensor, use_cache=True)
        cognition = out.logits[:, -1, :].float().unsqueeze(1)                               # [B,1,d]
        B = cognition.size(0)
        head = self.model.action_head
        noise = kwargs.get("noise")
        if noise is None:
            noise = torch.randn(B, self.config.chunk_size, self.config.action_dim, device=cognition.device,
                                dtype=cognition.dtype)
        if head.ddim_diffusion is None or head.ddim_diffusion.num_timesteps != num_ddim_steps:
            head.create_ddim(ddim_step=num_ddim_steps)
        if cfg_scale > 1.0:
            noise = torch.cat([noise, noise], 0)
            unc = self.store.w32("model.action_head.net.z_embedder.uncondition")          # [1,d]
            z = torch.cat([cognition, unc.unsqueeze(0).expand(B, 1, -1)], 0)
            model_kwargs = dict(z=z, cfg_scale=cfg_scale)
            sample_fn = head.net.forward_with_cfg
        else:
            model_kwargs = dict(z=cognition)
            sample_fn = head.net.forward
        res = head.ddim_diffusion.ddim_sample_loop(sample_fn, noise.shape, noise, clip_denoised=False,
                                                   model_kwargs=model_kwargs, eta=0.0, device=cognition.device,
                                                   return_trajectory=bool(kwargs.get("return_trajectory")))
        samples, traj = (res if isinstance(res, tuple) else (res, None))
        if cfg_scale > 1.0:
            samples = samples[:B]
        actions = self._denorm(samples[0].cpu().numpy(), action_norms).tolist()
        if traj is not None:
            return actions, samples, traj
        return actions
