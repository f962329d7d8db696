"""Vision-projector factory (mirror of dexbotic/model/modules/mm_projector/builder.py:36-81)."""
from __future__ import annotations

import re

import torch
import torch.nn as nn

from .... import _lib as L
from .... import functional as Fn
from typing import Optional

from ....engine import ParamStore, current_store

REQUIRED = ("mm_projector_type", "mm_hidden_size", "hidden_size")


class MlpProjector(nn.Module):
    """`mlpNx_gelu`: Linear -> (GELU(erf) -> Linear) x (N-1); state_dict names `0.weight`, `2.weight`, ..."""

    def __init__(self, store: ParamStore, prefix: str, depth: int, in_dim: int, out_dim: int):
        super().__init__()
        if depth < 1:
            raise ValueError("mlpNx_gelu needs N >= 1")
        self.store, self.p, self.depth = store, prefix, depth
        store.new_bucket()
        store.register([(prefix + "0.weight", (out_dim, in_dim)), (prefix + "0.bias", (out_dim,))])
        for i in range(1, depth):          # nn.Sequential indices: Linear at 0, 2, 4, ... with GELU modules in between
            store.register([(f"{prefix}{2 * i}.weight", (out_dim, out_dim)), (f"{prefix}{2 * i}.bias", (out_dim,))])

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        st, p = self.store, self.p
        anchor = st.params[p + "0.weight"]
        if self.depth == 1:
            return Fn.LinearFn.apply(x, anchor, st, p + "0.weight", p + "0.bias", L.ACT_NONE, None)
        if self.depth == 2:
            return Fn.MlpFn.apply(x, anchor, st, p + "0.weight", p + "0.bias", p + "2.weight", p + "2.bias",
                                  L.ACT_GELU_ERF)
        for i in range(self.depth):        # deeper stacks: GELU fused into every linear but the last
            act = L.ACT_GELU_ERF if i < self.depth - 1 else L.ACT_NONE
            x = Fn.LinearFn.apply(x, anchor, st, f"{p}{2 * i}.weight", f"{p}{2 * i}.bias", act, None)
        return x


class LinearProjector(nn.Module):
    """`linear` (bias) and `linearNx` (input width N x mm_hidden_size, bias only with config.projector_bias: builder.py:51-61)"""

    def __init__(self, store: ParamStore, prefix: str, in_dim: int, out_dim: int, bias: bool = True):
        super().__init__()
        self.store, self.p, self.has_bias = store, prefix, bias     # ('bias' itself is the parameter's attribute name)
        store.new_bucket()
        store.register([(prefix + "weight", (out_dim, in_dim))] + ([(prefix + "bias", (out_dim,))] if bias else []))

    def forward(self, x):
        st, p = self.store, self.p
        return Fn.LinearFn.apply(x, st.params[p + "weight"], st, p + "weight", p + "bias" if self.has_bias else None,
                                 L.ACT_NONE, None)


def build_vision_projector(config, store: Optional[ParamStore] = None, prefix: str = "model.mm_projector."):
    """reference signature ``build_vision_projector(config)``; the arena comes from the enclosing build context"""
    store = current_store(store)
    missing = [k for k in REQUIRED if not hasattr(config, k)]
    if missing:
        raise ValueError(f"Missing required config keys: {missing}")
    projector_type = getattr(config, "mm_projector_type", "mlp2x_gelu")
    if projector_type == "linear":
        return LinearProjector(store, prefix, config.mm_hidden_size, config.hidden_size)
    m = re.match(r"^mlp(\d+)x_gelu$", projector_type)
    if m:
        return MlpProjector(store, prefix, int(m.group(1)), config.mm_hidden_size, config.hidden_size)
    m = re.match(r"^linear(\d+)x$", projector_type)
    if m:
        return LinearProjector(store, prefix, config.mm_hidden_size * int(m.group(1)), config.hidden_size,
                               bias=bool(getattr(config, "projector_bias", False)))
    if projector_type == "mlp_downsample":
        raise NotImplementedError("projector 'mlp_downsample' (2x2 token merge + LayerNorm, used by the NaVILA family) is not on "
                                  "the CogACT / pi0 / MemVLA paths")
    raise ValueError(f"Unknown projector type: {projector_type}")
