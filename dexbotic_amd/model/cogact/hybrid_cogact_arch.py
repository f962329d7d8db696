"""HybridCogACT: text cross-entropy + diffusion action loss co-training — host-side mirror of
dexbotic/model/cogact/hybrid_cogact_arch.py:52-207 on libdexbotic_amd kernels.

One VLM prefill feeds both heads: ``lm_head`` logits scored against the labels of the samples that carry text
(``has_text``) with the HF causal-LM cross-entropy, and the DiT action head on the cognition token with a
``has_action``-weighted mean of the per-sample eps-MSE.  loss = text_loss + action_loss.
"""
from __future__ import annotations

import numpy as np
import torch

from ... import functional as Fn
from ...constants import IGNORE_INDEX
from ..dexbotic_arch import CausalLMOutputDexbotic
from .cogact_arch import CogActConfig, CogACTForCausalLM


class HybridCogACTForCausalLM(CogACTForCausalLM):
    config_class = CogActConfig
    # the text loss is a mean over the batch's labelled tokens: the mean over a merged batch is not the mean of the micro-batch
    # means when their token counts differ
    coalescible_micro_batches = False

    def unused_parameter_names(self):
        """lm_head trains here; the rest as in CogACT"""
        return [n for n in super().unused_parameter_names() if n != "lm_head.weight"]

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, inputs_embeds=None,
                labels=None, use_cache=None, output_attentions=None, output_hidden_states=None, images=None,
                return_dict=None, cache_position=None, actions=None, states=None, repeated_diffusion_steps: int = 4,
                has_action=None, has_text=None, **kwargs) -> CausalLMOutputDexbotic:
        (_, position_ids, attention_mask, past_key_values, inputs_embeds, labels_t, cache_position
         ) = self.model._prepare_inputs_labels_for_multimodal(input_ids, position_ids, attention_mask, past_key_values,
                                                               labels, cache_position, images)
        hidden = self.model.run_llm(inputs_embeds, attention_mask)                       # [B,S,d]
        B, S, d = hidden.shape
        plan = self.model._last_plan
        text_loss = action_loss = None
        if labels is not None:
            assert has_action is not None, "has_action must be provided"
            ht = np.asarray(has_text.cpu() if torch.is_tensor(has_text) else has_text).astype(bool).reshape(-1)
            lab = plan.labels.copy()
            if not ht.any():
                lab[~ht] = IGNORE_INDEX            # no text anywhere: score nothing, weight 0 (hybrid_cogact_arch.py:133-143)
            shifted = np.full_like(lab, IGNORE_INDEX)
            shifted[:, :-1] = lab[:, 1:]
            n_valid = int((shifted != IGNORE_INDEX).sum())
            text_loss, _ = Fn.LmHeadLossFn.apply(hidden, self.store.params["lm_head.weight"], self.store, "lm_head.weight",
                                                 torch.from_numpy(shifted.reshape(-1)).to(hidden.device), n_valid)
            text_loss = text_loss * float(ht.any())
        if attention_mask is not None and actions is not None:
            assert has_text is not None, "has_text must be provided"
            idx = torch.from_numpy(np.arange(B, dtype=np.int64) * S + plan.last_index).to(hidden.device)
            cognition = Fn.GatherRowsFn.apply(hidden.reshape(B * S, d), idx)                   # [B,d] fp32
            A, T = self.config.action_dim, self.config.chunk_size
            acts = actions.reshape(actions.size(0), -1, A).float()[:, :T, :]
            R = repeated_diffusion_steps
            ha = torch.as_tensor(has_action).to(hidden.device).reshape(-1).float().repeat(R)
            action_loss = self.model.action_head_module.loss(
                acts.repeat(R, 1, 1), cognition.repeat(R, 1).unsqueeze(1), reduction="none", sample_weight=ha,
                noise=kwargs.get("noise"), timestep=kwargs.get("timesteps"), drop_ids=kwargs.get("drop_ids"))
        loss = None
        if text_loss is not None and action_loss is not None:
            loss = text_loss + action_loss
        elif text_loss is not None:
            loss = text_loss
        elif action_loss is not None:
            loss = action_loss
        return CausalLMOutputDexbotic(loss=loss, text_loss=text_loss, action_loss=action_loss, logits=hidden,
                                      hidden_states=(hidden,))
