"""Action-model factory (mirror of dexbotic/model/cogact/action_model/builder.py:5-27)."""
from __future__ import annotations

from typing import Optional

from ....engine import ParamStore, current_store
from .action_models import ActionModel, LinearModel

REQUIRED = ("action_model_type", "hidden_size", "action_dim", "chunk_size")


def build_action_model(config, store: Optional[ParamStore] = None, prefix: str = "model.action_head."):
    """reference signature ``build_action_model(config)``; the arena comes from the enclosing build context"""
    store = current_store(store)
    missing = [k for k in REQUIRED if not hasattr(config, k)]
    if missing:                                     # exp/utils.py:43-52 require_config_keys
        raise ValueError(f"Missing required config keys: {missing}")
    model_type = config.action_model_type
    if "Linear" in model_type:
        return LinearModel(store=store, prefix=prefix, model_type=model_type, token_size=config.hidden_size,
                           in_channels=config.action_dim, future_action_window_size=config.chunk_size - 1,
                           past_action_window_size=0)
    if "DiT" in model_type:
        # memvla/action_model/builder.py:14-20: a config that carries per_token_size builds the per-attention DiT
        pts = getattr(config, "per_token_size", None)
        return ActionModel(store=store, prefix=prefix, model_type=model_type, token_size=config.hidden_size,
                           in_channels=config.action_dim, future_action_window_size=config.chunk_size - 1,
                           past_action_window_size=0, use_per_attn=pts is not None, per_token_size=pts)
    raise ValueError(f"Unknown action model type: {model_type}")
