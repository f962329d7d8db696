"""Discrete-token VLA policy: host-side mirror of dexbotic/model/discrete_vla/discrete_vla_arch.py.

``DiscreteVLAForCausalLM.inference_action`` (:14-50): generate digit tokens after the prompt until the stop string,
decode them, take the first 7 integers as action bins, map bins back to [-1, 1] (:52-58, integer arithmetic —
bit-exact contract) and de-normalise.  The decoder pass, lm_head and token choice run on
libdexbotic_amd kernels through ``DexboticForCausalLM.generate``; tokenizer and conversation template stay the
caller's objects, exactly as in the reference.
"""
from __future__ import annotations

import re

import numpy as np
import torch

from ..dexbotic_arch import ActionOutputForCausalLM, DexboticConfig, DexboticForCausalLM


class KeywordsStoppingCriteria:
    """stop when the decoded tail of the continuation contains a keyword (tokenization/conversation.py:15-48)"""

    def __init__(self, keywords, tokenizer, input_ids):
        self.keywords = list(keywords)
        self.keyword_ids = []
        self.max_keyword_len = 0
        for kw in self.keywords:
            ids = list(tokenizer(kw).input_ids)
            if len(ids) > 1 and ids[0] == getattr(tokenizer, "bos_token_id", None):
                ids = ids[1:]
            self.max_keyword_len = max(self.max_keyword_len, len(ids))
            self.keyword_ids.append(torch.tensor(ids))
        self.tokenizer = tokenizer
        self.start_len = input_ids.shape[1]

    def _one(self, output_ids: torch.Tensor) -> bool:
        offset = min(output_ids.shape[1] - self.start_len, self.max_keyword_len)
        for kid in self.keyword_ids:
            if output_ids.shape[1] >= kid.shape[0] and torch.equal(output_ids[0, -kid.shape[0]:].cpu(), kid):
                return True
        text = self.tokenizer.batch_decode(output_ids[:, -offset:], skip_special_tokens=True)[0] if offset > 0 else ""
        return any(kw in text for kw in self.keywords)

    def __call__(self, output_ids: torch.Tensor, scores=None, **kwargs) -> bool:
        return all(self._one(output_ids[i:i + 1]) for i in range(output_ids.shape[0]))


class DiscreteVLAForCausalLM(DexboticForCausalLM, ActionOutputForCausalLM):
    config_class = DexboticConfig

    def inference_action(self, input_ids, image_tensor, inference_args={}, **kwargs):
        attempt = 0
        while attempt < 40:                                      # discrete_vla_arch.py:15-22
            try:
                return self._real_inference_action(input_ids, image_tensor, inference_args, **kwargs)
            except (ValueError, IndexError) as e:                # an unparsable sample: draw again
                attempt += 1
                print(f"Attempt {attempt} failed: {e}")

    def _real_inference_action(self, input_ids, image_tensor, inference_args={}, **kwargs):
        conv = inference_args.get("conv")
        tokenizer = inference_args.get("tokenizer")
        vocab_size = inference_args.get("vocab_size")
        action_norms = inference_args.get("action_norms")
        two = getattr(getattr(conv, "sep_style", None), "name", "") == "TWO"
        stop_str = conv.sep2 if two else conv.sep
        criteria = KeywordsStoppingCriteria([stop_str], tokenizer, input_ids)
        out = self.generate(input_ids, images=image_tensor,
                            max_new_tokens=inference_args.get("max_new_tokens", 1024),
                            do_sample=inference_args.get("do_sample", True),
                            temperature=inference_args.get("temperature", 0.7),
                            return_dict_in_generate=True, stopping_criteria=[criteria],
                            generator=kwargs.get("generator"))
        new = out.sequences[0, input_ids.shape[1]:]
        text = tokenizer.decode(new, skip_special_tokens=False).strip(stop_str)
        actions = self._discrete_action_to_continuous(text, vocab_size)
        return self._denorm(actions, action_norms).tolist()

    def _discrete_action_to_continuous(self, action_str: str, vocab_size: int):
        """bins [0, vocab_size-1] -> [-1, 1] (discrete_vla_arch.py:52-58)"""
        actions = re.findall(r"\d+", action_str)[:7]
        actions = np.array([int(a) for a in actions], dtype=np.float32).reshape(1, -1)
        return (actions / (vocab_size - 1)) * 2 - 1
