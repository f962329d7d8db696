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

    def dev(self, device) -> dict:
        """device copies of the integer arrays the kernels consume, uploaded once per plan (a cached plan re-used by
        the next step costs no host->device traffic at all).  A fresh plan — every step of a real fine-tune has new token
        ids — costs ONE asynchronous copy: the six arrays are packed into one pinned host buffer and sent with a single
        non-blocking transfer on the current stream (pageable sources would make each of six small copies wait for the
        stream to drain), the device tensors are typed views of that one allocation."""
        import torch
        cache = self.__dict__.setdefault("_dev", {})
        key = str(device)
        ent = cache.get(key)
        if ent is None:
            B, S = self.plan.shape
            parts = (("plan", self.plan.reshape(-1), np.int64), ("labels", self.labels.reshape(-1), np.int64),
                     ("last_flat", np.arange(B, dtype=np.int64) * S + self.last_index, np.int64),
                     ("kv_start", self.kv_start, np.int32), ("kv_end", self.kv_end, np.int32),
                     ("mask", self.attention_mask.reshape(-1), np.bool_))
            offs, total = [], 0
            for _, a, dt in parts:
                offs.append(total)
                total += (a.size * np.dtype(dt).itemsize + 15) // 16 * 16
            cuda = torch.device(device).type == "cuda"
            host = torch.empty(max(total, 16), dtype=torch.uint8, pin_memory=cuda)
            hv = host.numpy()
            for (_, a, dt), o in zip(parts, offs):
                n = a.size * np.dtype(dt).itemsize
                hv[o:o + n] = np.ascontiguousarray(a, dtype=dt).view(np.uint8).reshape(-1)
            buf = host.to(device, non_blocking=True)
            tdt = {np.int64: torch.int64, np.int32: torch.int32, np.bool_: torch.bool}
            ent = {"_host": host, "_buf": buf}
            for (name, a, dt), o in zip(parts, offs):
                n = a.size * np.dtype(dt).itemsize
                ent[name] = buf[o:o + n].view(tdt[dt])
            ent["labels"] = ent["labels"].view(B, S)
            ent["mask"] = ent["mask"].view(B, S)
            cache[key] = ent
        return ent


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


class PlanCache:
    """Splice plans keyed by the CONTENT of the integer inputs (ids / mask / labels are a few KB), plus an identity
    shortcut for device-resident id tensors so that a batch object seen before costs no device->host copy (the
    reference pays ~10 tiny kernels and several syncs per sample here, dexbotic_arch.py:182-373)."""

    def __init__(self, capacity: int = 32):
        self.capacity = capacity
        self._by_content: dict = {}
        self._by_ident: dict = {}

    @staticmethod
    def _ident(t):
        import weakref
        return (id(t), t._version, tuple(t.shape)), weakref.ref(t)

    def host(self, t):
        """numpy view of an integer input; device tensors are copied (a stream sync) unless this very tensor object,
        unmodified, was seen before"""
        if t is None or isinstance(t, np.ndarray):
            return t
        if not t.is_cuda:
            return t.detach().numpy()
        key, ref = self._ident(t)
        hit = self._by_ident.get(key)
        if hit is not None and hit[0]() is t:
            return hit[1]
        arr = t.detach().cpu().numpy()
        if len(self._by_ident) >= 4 * self.capacity:
            self._by_ident.clear()
        self._by_ident[key] = (ref, arr)
        return arr

    def get(self, ids, mask, labels, n_img_rows: int, max_length, padding_side: str) -> SplicePlan:
        ids, mask, labels = self.host(ids), self.host(mask), self.host(labels)
        key = (ids.shape, ids.tobytes(), None if mask is None else mask.tobytes(),
               None if labels is None else labels.tobytes(), n_img_rows, max_length, padding_side)
        plan = self._by_content.get(key)
        if plan is None:
            plan = build_splice_plan(ids, mask, labels, n_img_rows, max_length, padding_side)
            if len(self._by_content) >= self.capacity:
                self._by_content.pop(next(iter(self._by_content)))
            self._by_content[key] = plan
        return plan
