"""Vision-tower factory (mirror of dexbotic/model/modules/mm_vision/builder.py:9-34)."""
from __future__ import annotations

from ....engine import ParamStore
from .clip.clip_encoder import CLIPVisionConfig, CLIPVisionTower
from .siglip.siglip_encoder import SiglipVisionConfig, SiglipVisionTower


def build_vision_tower(mm_vision_tower, store: ParamStore, prefix: str = "model.mm_vision_tower.", **kwargs):
    """`mm_vision_tower`: a checkpoint directory / hub name (selected on the substrings 'sig' / 'clip' / 'pe'
    exactly like the reference) or a CLIPVisionConfig (synthetic-weight benchmarks)."""
    vt = mm_vision_tower
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
