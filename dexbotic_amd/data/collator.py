"""Batch collation: dexbotic/data/collator.py:10-67 (right-padding, truncation, attention mask, key mapping)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Dict, Sequence

import torch

from ..constants import IGNORE_INDEX

_EOS_MARK = -300
_KEYS = {"image": "images", "actions": "actions", "action": "actions", "state": "states", "reward": "reward",
         "image_masks": "image_masks", "has_action": "has_action", "has_text": "has_text"}


def _pad_rows(rows, value):
    n = max(int(r.shape[0]) for r in rows)
    out = rows[0].new_full((len(rows), n), value)
    for i, r in enumerate(rows):
        out[i, :r.shape[0]] = r
    return out


@dataclass
class DataCollatorForSupervisedDataset:
    tokenizer: Any

    def __call__(self, instances: Sequence[Dict]) -> Dict[str, torch.Tensor]:
        tok = self.tokenizer
        ids = [inst["input_ids"] for inst in instances]
        labels = [inst["labels"] for inst in instances]
        shared = tok.pad_token_id == tok.eos_token_id
        if shared:        # genuine eos tokens must stay visible in the mask (collator.py:21-23, in place like the reference)
            for r in ids:
                r[r == tok.eos_token_id] = _EOS_MARK
        input_ids = _pad_rows(ids, tok.pad_token_id)[:, :tok.model_max_length]
        labels = _pad_rows(labels, IGNORE_INDEX)[:, :tok.model_max_length]
        attention_mask = input_ids.ne(tok.pad_token_id)
        if shared:
            input_ids[input_ids == _EOS_MARK] = tok.eos_token_id
        batch = dict(input_ids=input_ids, labels=labels, attention_mask=attention_mask)
        for key, name in _KEYS.items():
            if key in instances[0]:
                vals = [inst[key] for inst in instances]
                same = all(v is not None and v.shape == vals[0].shape for v in vals)
                batch[name] = torch.stack(vals) if same else vals
        return batch
