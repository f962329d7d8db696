"""HIP-graph replay of launch-bound inference loops.

The action samplers of the secondary policies are long chains of tiny launches — MemVLA's DiT-L with perceptual attention is
~480 kernels per DDIM step, pi0's KV-cached Euler step ~250 — issued from Python at ~10 us apiece while the kernels themselves
take a few microseconds: the request is host-bound.  ``GraphCache.run(key, fn, inputs)`` runs ``fn(**inputs)`` eagerly the first
time a key is seen (every lazily created resource — split-K scratch of the stream, device tables — comes into being), captures
it into a HIP graph the second time and replays it afterwards with the inputs copied into the captured buffers.  ``fn`` must be
free of host synchronisation and host->device copies (tensors in, tensors out).  One private stream per cache: the library
keeps per-stream scratch for the life of the process.

(The DB-CogACT request is GPU-bound — its graph path, cogact_arch._graph_sample, is off by default.)
"""
from __future__ import annotations

import contextlib
import gc
import os
from typing import Callable, Dict, Hashable

import torch


@contextlib.contextmanager
def capture(graph: "torch.cuda.CUDAGraph", stream: "torch.cuda.Stream"):
    """``torch.cuda.graph(graph, stream=stream)`` with Python's cyclic garbage collector held off for the duration of the capture.
    torch collects once BEFORE the capture begins; a generation-0 collection that the capture's own allocations trigger in the middle
    of it finalises whatever cyclic garbage earlier code left behind (models, trainers with their side streams and events, older graphs)
    — destructors that talk to the HIP runtime while a stream is capturing (global capture mode) abort the process (seen once in the GPU
    suite: `Fatal Python error: Aborted ... Garbage-collecting` inside a captured request, round 6)."""
    was_on = gc.isenabled()
    gc.disable()
    try:
        with torch.cuda.graph(graph, stream=stream):
            yield
    finally:
        if was_on:
            gc.enable()


def enabled(default: bool = True) -> bool:
    v = os.environ.get("DXA_INFER_GRAPH")
    return default if v is None else v != "0"


class GraphCache:
    def __init__(self, device, capacity: int = 8):
        self.device = torch.device(device)
        self.capacity = capacity
        self.entries: Dict[Hashable, dict] = {}
        self.stream = None

    def clear(self) -> None:
        self.entries.clear()

    def run(self, key: Hashable, fn: Callable[..., torch.Tensor], inputs: Dict[str, torch.Tensor]) -> torch.Tensor:
        """``fn(**inputs) -> tensor``; the result lives in graph-owned memory and is valid until the next run() of this key"""
        if self.device.type != "cuda":
            return fn(**inputs)
        if self.stream is None:
            self.stream = torch.cuda.Stream(device=self.device)
        key = (key, tuple((k, tuple(v.shape), v.dtype) for k, v in sorted(inputs.items())))
        ent = self.entries.get(key)
        cur = torch.cuda.current_stream(self.device)
        if ent is None:
            if len(self.entries) >= self.capacity:
                self.entries.pop(next(iter(self.entries)))
            ent = self.entries[key] = {"static": {k: v.clone() for k, v in inputs.items()}, "graph": None, "out": None}
            self.stream.wait_stream(cur)
            with torch.cuda.stream(self.stream):
                out = fn(**ent["static"])
            cur.wait_stream(self.stream)
            return out
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            for k, v in inputs.items():
                ent["static"][k].copy_(v, non_blocking=True)
            if ent["graph"] is None:
                g = torch.cuda.CUDAGraph()
                with capture(g, self.stream):
                    ent["out"] = fn(**ent["static"])
                ent["graph"] = g
            ent["graph"].replay()
        cur.wait_stream(self.stream)
        return ent["out"]
