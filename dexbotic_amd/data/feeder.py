"""Host batches -> device, one step ahead of the compute stream.

The reference leaves this to HF ``Trainer`` (``_prepare_inputs``: ``tensor.to(device)`` of the collator's output,
``dataloader_pin_memory=True``).  Here the step itself never waits for the host:

  * floating-point inputs (``images``, ``actions``, ``states`` ...) are staged in pinned host memory and uploaded with
    non-blocking copies on a dedicated COPY stream while the previous step computes; the compute stream waits for the
    batch's event only, and ``record_stream`` keeps the allocator from recycling the buffers early;
  * integer inputs (``input_ids``, ``attention_mask``, ``labels``) stay ON THE HOST: the multimodal splice plan is host
    arithmetic over a few KB (splice.build_splice_plan) and its device form is one packed pinned upload
    (splice.SplicePlan.dev) — handing device ids to the model would cost a device->host copy, i.e. a stream drain per step.

``DeviceFeeder(iterable_of_collated_batches, device)`` is an iterator of ready batches for ``NativeTrainer.step`` /
``model(**batch)``.
"""
from __future__ import annotations

from typing import Dict, Iterable, Iterator, Optional

import torch

HOST_KEYS = ("input_ids", "attention_mask", "labels")


class DeviceFeeder:
    def __init__(self, batches: Iterable[Dict], device, depth: int = 1, host_keys=HOST_KEYS):
        self.it: Iterator[Dict] = iter(batches)
        self.device = torch.device(device)
        self.host_keys = set(host_keys)
        self.depth = max(1, int(depth))
        self.copy_stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None
        self._queue = []            # [(batch on device, event)]
        self._pinned: Dict[tuple, list] = {}     # (key, shape, dtype) -> ring of pinned staging buffers
        self._ring = 0
        self._done = False
        self._slot_events: Dict[tuple, "torch.cuda.Event"] = {}
        self._staged: list = []

    def _stage(self, key: str, t: torch.Tensor) -> torch.Tensor:
        if t.is_pinned():
            return t
        k = (key, tuple(t.shape), t.dtype)
        ring = self._pinned.get(k)
        if ring is None:
            ring = self._pinned[k] = [torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for _ in range(self.depth + 2)]
        slot = self._ring % len(ring)
        ev = self._slot_events.get((k, slot))
        if ev is not None:
            ev.synchronize()          # the upload that last read this staging buffer (long finished; a host-side formality)
        buf = ring[slot]
        buf.copy_(t)
        self._staged.append((k, slot))
        return buf

    def _prefetch(self) -> None:
        if self._done:
            return
        try:
            b = next(self.it)
        except StopIteration:
            self._done = True
            return
        out, ev = {}, None
        if self.copy_stream is None:
            self._queue.append((dict(b), None))
            return
        with torch.cuda.stream(self.copy_stream):
            for k, v in b.items():
                if torch.is_tensor(v) and k not in self.host_keys and not v.is_cuda:
                    out[k] = self._stage(k, v).to(self.device, non_blocking=True)
                else:
                    out[k] = v
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        for ks in self._staged:
            self._slot_events[ks] = ev
        self._staged = []
        self._ring += 1
        self._queue.append((out, ev))

    def __iter__(self):
        return self

    def __next__(self) -> Dict:
        while len(self._queue) < self.depth + 1 and not self._done:
            self._prefetch()
        if not self._queue:
            raise StopIteration
        batch, ev = self._queue.pop(0)
        if ev is not None:
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)
            for v in batch.values():
                if torch.is_tensor(v) and v.is_cuda:
                    v.record_stream(cur)
        return batch
