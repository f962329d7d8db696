"""Action label transforms of the discrete / hybrid policies (encode direction).

Mirror of ``ActionNormAnd2String`` (dexbotic/data/dataset/transform/action.py:283-397): same constructor, same
``__call__(episode_data_dict)`` contract, same integer results.  Host numpy on purpose: the row is a few hundred
float64 values per episode and its contract is BIT-EXACT integers — ``np.round`` (round-half-to-even) decides bins on
exact halves, so the arithmetic is kept in float64 numpy exactly as the reference evaluates it; the inverse
(digit tokens -> bins -> [-1,1] -> physical units) is DiscreteVLAForCausalLM._discrete_action_to_continuous and
ActionOutputForCausalLM._denorm (model/dexbotic_arch.py)."""
from __future__ import annotations

import copy
from typing import List

import numpy as np


class ActionNormAnd2String:
    """normalise ``action`` [T, D] to [-1, 1] with per-dataset / per-prompt min-max statistics, bin it to
    ``vocab_size`` levels and render one string per step (the text target of DiscreteVLA / HybridCogACT);
    sets ``episode_data_dict['action']`` and, unless present, ``['answer']``."""

    def __init__(self, statistic_mapping: dict = {"default": {"min": -1, "max": 1}}, vocab_size: int = 255,
                 string_format: str = " {value}", add_answer: bool = True):
        if "default" not in statistic_mapping:
            raise AssertionError("the default statistic mapping should be provided")
        self.vocab_size = vocab_size
        self.statistic_mapping = statistic_mapping
        self.string_format = string_format
        self.add_answer = add_answer

    def _stats_for(self, dataset, prompt, action_dim: int, episode: dict) -> dict:
        sm = self.statistic_mapping
        if dataset not in sm:
            st = copy.deepcopy(sm["default"])
        elif prompt not in sm[dataset]:
            st = copy.deepcopy(sm[dataset]["default"])
        else:
            st = copy.deepcopy(sm[dataset][prompt])
        if isinstance(st["min"], (int, float)):
            st["min"], st["max"] = [st["min"]], [st["max"]]
        if len(st["min"]) == 1:
            st["min"] = np.array(list(st["min"]) * action_dim)
            st["max"] = np.array(list(st["max"]) * action_dim)
        if "trajectory" in episode:                       # chunked targets: the statistics repeat per predicted step
            n = episode["meta_data"]["trajectory_length"]
            st["min"] = np.concatenate([st["min"] for _ in range(n)], axis=0)
            st["max"] = np.concatenate([st["max"] for _ in range(n)], axis=0)
        return st

    def __call__(self, episode_data_dict: dict, **kwargs) -> dict:
        if "action" not in episode_data_dict:
            return episode_data_dict
        action = episode_data_dict["action"]
        prompt = episode_data_dict["prompt"][0]           # one prompt per episode
        dataset = episode_data_dict["meta_data"]["dataset"]
        st = self._stats_for(dataset, prompt, len(action[0]), episode_data_dict)
        normed = self._norm_action(action, np.asarray(st["min"]), np.asarray(st["max"]))
        episode_data_dict["action"] = normed
        strings = self._bin2string(self._action2bin(normed, self.vocab_size), self.string_format)
        if self.add_answer and "answer" not in episode_data_dict:
            episode_data_dict["answer"] = strings
        return episode_data_dict

    def _norm_action(self, action, min, max) -> np.ndarray:
        """clip to [min, max], then (a - min) / (max - min + 1e-8) * 2 - 1   (:378-384)"""
        lo, hi = min.reshape(1, -1), max.reshape(1, -1)
        a = np.clip(action, lo, hi)
        return (a - lo) / (hi - lo + 1e-8) * 2 - 1

    def _action2bin(self, action, vocab_size) -> np.ndarray:
        """round((a + 1) / 2 * (V - 1)) with numpy's round-half-to-even, clipped to [0, V - 1]   (:386-390)"""
        b = np.round((action + 1) / 2 * (vocab_size - 1))
        return np.clip(b, 0, vocab_size - 1)

    def _bin2string(self, action, string_format) -> List[str]:
        """[T, D] bins -> T strings, every value through ``string_format``   (:392-397)"""
        return ["".join(string_format.format(value=int(v)) for v in row) for row in action]
