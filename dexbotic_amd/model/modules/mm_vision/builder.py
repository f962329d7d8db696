"""Vision-tower factory (mirror of dexbotic/model/modules/mm_vision/builder.py:9-34)."""
from __future__ import annotations

from typing import Optional

from ....engine import ParamStore, current_store
from .clip.clip_encoder import CLIPVisionConfig, CLIPVisionTower
from .siglip.siglip_encoder import SiglipVisionConfig, SiglipVisionTower


def build_vision_tower(mm_vision_tower, store: Optional[ParamStore] = None, prefix: str = "model.mm_vision_tower.",
                       **kwargs):
    """Reference signature ``build_vision_tower(mm_vision_tower_cfg, **kwargs)``: `mm_vision_tower` is a checkpoint
    directory / hub name (selected on the substrings 'sig' / 'clip' / 'pe' exactly like the reference), a config object
    carrying ``mm_vision_tower`` (mm_vision/builder.py:10), or a CLIPVisionConfig / SiglipVisionConfig (synthetic-weight
    benchmarks).  The arena comes from the enclosing build context (engine.building) unless passed explicitly."""
    store = current_store(store)
    vt = getattr(mm_vision_tower, "mm_vision_tower", mm_vision_tower)
    if isinstance(vt, CLIPVisionConfig):
        return CLIPVisionTower(vt, store, prefix, **kwargs)
    if isinstance(vt, SiglipVisionConfig):
        return SiglipVisionTower(vt, store, prefix, **kwargs)
    if isinstance(vt, dict):
        if "siglip" in str(vt.get("model_type", "")):
            return SiglipVisionTower(SiglipVisionConfig.from_any(vt), store, prefix, **kwargs)
        return CLIPVisionTower(CLIPVisionConfig.from_any(vt), store, prefix, **kwargs)
    if isinstance(vt, str):
        low = vt.lower()
        if "sig" in low:
            return SiglipVisionTower(vt, store, prefix, **kwargs)
        if "clip" in low:
            return CLIPVisionTower(vt, store, prefix, **kwargs)
        if "pe" in low:
            raise NotImplementedError("PEVisionTower is outside the north-star path (SURVEY.md §2 row 2)")
    raise ValueError(f"Unknown vision tower: {vt}")
