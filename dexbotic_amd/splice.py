"""Host-side integer plan of the multimodal splice.

The reference builds ``inputs_embeds`` with a Python loop over the batch of ``torch.cat/split/zeros``
calls (dexbotic/model/dexbotic_arch.py:182-373, ~10 tiny kernels per sample, SURVEY.md §3 hot loop #2).
Here the *indices* are computed once per batch on the host and ONE kernel (dxa_splice_fwd) gathers
embedding rows / image-feature rows / zero padding.  Semantics kept exactly:
  * padding removed by ``attention_mask`` first (:219-222);
  * every IMAGE_TOKEN_INDEX placeholder is replaced by the next ``image_features[cur_image_idx]`` block
    (all V*N_v rows of that sample), ``cur_image_idx`` running globally over the batch (:229-233,:291-304);
    a sample without a placeholder still consumes one block (:264-271);
  * truncation to ``tokenizer_model_max_length`` (:238-243); right (default) or left padding (:342-371);
  * new attention mask / labels (IGNORE_INDEX on image rows and padding).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np

from .constants import IGNORE_INDEX, IMAGE_TOKEN_INDEX

PLAN_PAD = np.iinfo(np.int64).min


@dataclass
class SplicePlan:
    plan: np.ndarray            # [B, S] int64: >=0 token id, <=-1 image row (-1-k), PLAN_PAD padding
    attention_mask: np.ndarray  # [B, S] bool
    labels: np.ndarray          # [B, S] int64
    lengths: np.ndarray         # [B]
    kv_start: np.ndarray        # [B] int32 first valid key
    kv_end: np.ndarray          # [B] int32 one past the last valid key
    last_index: np.ndarray      # [B] int64 position of the last un-padded token (cognition token)


def build_splice_plan(input_ids: np.ndarray, attention_mask: Optional[np.ndarray], labels: Optional[np.ndarray],
                      n_img_rows: int, max_length: Optional[int] = None, padding_side: str = "right") -> SplicePlan:
    input_ids = np.asarray(input_ids, dtype=np.int64)
    B, Lt = input_ids.shape
    mask = np.ones((B, Lt), dtype=bool) if attention_mask is None else np.asarray(attention_mask).astype(bool)
    lab = np.full((B, Lt), IGNORE_INDEX, dtype=np.int64) if labels is None else np.asarray(labels, dtype=np.int64)
    rows, row_labels = [], []
    block = 0
    img_rows = -1 - np.arange(n_img_rows, dtype=np.int64)
    for b in range(B):
        ids, lb = input_ids[b][mask[b]], lab[b][mask[b]]
        where = np.flatnonzero(ids == IMAGE_TOKEN_INDEX)
        if where.size == 0:
            rows.append(ids)
            row_labels.append(lb)
            block += 1
            continue
        pieces, lpieces, prev = [], [], 0
        for pos in where:
            pieces += [ids[prev:pos], img_rows - block * n_img_rows]
            lpieces += [lb[prev:pos], np.full(n_img_rows, IGNORE_INDEX, dtype=np.int64)]
            block += 1
            prev = pos + 1
        pieces.append(ids[prev:])
        lpieces.append(lb[prev:])
        rows.append(np.concatenate(pieces))
        row_labels.append(np.concatenate(lpieces))
    if max_length is not None:
        rows = [r[:max_length] for r in rows]
        row_labels = [r[:max_length] for r in row_labels]
    lengths = np.array([len(r) for r in rows], dtype=np.int64)
    S = int(lengths.max()) if B else 0
    plan = np.full((B, S), PLAN_PAD, dtype=np.int64)
    new_mask = np.zeros((B, S), dtype=bool)
    new_labels = np.full((B, S), IGNORE_INDEX, dtype=np.int64)
    kv_start = np.zeros(B, dtype=np.int32)
    kv_end = np.zeros(B, dtype=np.int32)
    for b, (r, l) in enumerate(zip(rows, row_labels)):
        n = len(r)
        lo = S - n if padding_side == "left" else 0
        plan[b, lo:lo + n] = r
        new_mask[b, lo:lo + n] = True
        new_labels[b, lo:lo + n] = l
        kv_start[b], kv_end[b] = lo, lo + n
    # cogact_arch.py:110-120: first index where cumsum(mask) reaches its maximum
    cs = new_mask.cumsum(axis=1)
    last = (cs == cs.max(axis=1, keepdims=True)).argmax(axis=1).astype(np.int64) if S else np.zeros(B, np.int64)
    return SplicePlan(plan, new_mask, new_labels, lengths, kv_start, kv_end, last)
