"""Device-side PreprocessRGB — drop-in for dexbotic/data/dataset/rgb_preprocess.py:5-44.

The reference pads the PIL frame to a square, then lets the HF CLIP image processor resize (Pillow bicubic), crop,
rescale and normalise it on the host, one frame at a time.  Here the uint8 frame goes to the MI355X as it is
(h*w*3 bytes over PCIe instead of 3*224*224 floats) and libdexbotic_amd's dxa_image_preprocess does all of that in two
launches, bit-exact with Pillow's 8-bit resampler (tests/test_image_gpu.py).  Same constructor, same call."""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np
import torch

from ... import kernels as K


def _edge(v, key):
    """one entry of the size / crop_size of an HF image processor (dict, SizeDict or a bare int)"""
    if isinstance(v, int):
        return v
    if isinstance(v, dict):
        return v.get(key)
    return getattr(v, key, None)


class ImageProcessorSpec:
    """the fields of an HF CLIPImageProcessor this path reads (defaults: openai/clip-vit-large-patch14)"""

    def __init__(self, size: int = 224, crop_size: int = 224, image_mean: Sequence[float] = (0.48145466, 0.4578275, 0.40821073),
                 image_std: Sequence[float] = (0.26862954, 0.26130258, 0.27577711), rescale_factor: float = 1 / 255):
        self.size = {"shortest_edge": size}
        self.crop_size = {"height": crop_size, "width": crop_size}
        self.image_mean, self.image_std, self.rescale_factor = tuple(image_mean), tuple(image_std), rescale_factor


def to_uint8_hwc(image) -> torch.Tensor:
    """PIL.Image / numpy / torch frame -> contiguous uint8 [h, w, 3] tensor (host or device, unchanged)"""
    if isinstance(image, torch.Tensor):
        t = image
    elif isinstance(image, np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(image))
    else:                                               # PIL image: the reference feeds .convert('RGB') frames
        t = torch.from_numpy(np.asarray(image.convert("RGB") if getattr(image, "mode", "RGB") != "RGB" else image).copy())
    if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[-1] != 3:
        raise ValueError(f"expected a uint8 RGB frame [h, w, 3], got {t.dtype} {tuple(t.shape)}")
    return t.contiguous()


class PreprocessRGB:
    def __init__(self, image_processor, image_aspect_ratio=None, augmentations=None, image_pad_mode="mean",
                 device: Optional[torch.device] = None, dtype: torch.dtype = torch.float32):
        self.image_processor = image_processor
        self.image_aspect_ratio = image_aspect_ratio
        self.augmentations = augmentations
        self.image_pad_mode = image_pad_mode
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.dtype = dtype

    # ---- geometry / constants read off the processor exactly where the reference reads them
    def _crop(self):
        cs = getattr(self.image_processor, "crop_size", None) or self.image_processor.size
        return int(_edge(cs, "height")), int(_edge(cs, "width"))

    def _short(self):
        s = self.image_processor.size
        v = _edge(s, "shortest_edge")
        return int(v if v is not None else _edge(s, "height"))

    def _bg(self):
        if self.image_pad_mode == "zero":                                   # rgb_preprocess.py:20-21
            return (0, 0, 0)
        return tuple(int(x * 255) for x in self.image_processor.image_mean)  # rgb_preprocess.py:23

    def batch(self, frames: torch.Tensor, want_u8: bool = False):
        """frames: uint8 [n, h, w, 3] (host or device) -> [n, 3, H, W] on the device"""
        frames = frames.to(self.device, non_blocking=True)
        p = self.image_processor
        return K.image_preprocess(frames, pad=self.image_aspect_ratio == "pad", bg=self._bg(), size=self._short(),
                                  crop=self._crop(), mean=p.image_mean, std=p.image_std,
                                  rescale=getattr(p, "rescale_factor", 1 / 255), out_dtype=self.dtype, want_u8=want_u8)

    def __call__(self, image) -> torch.Tensor:
        if image is None:                                                   # rgb_preprocess.py:14-19
            ch, cw = self._crop()
            return torch.zeros(3, ch, cw, device=self.device, dtype=self.dtype)
        if self.augmentations:                                              # PIL-level augmentations stay on the host
            if self.image_aspect_ratio == "pad":
                from PIL import Image
                w, h = image.size
                if w != h:
                    side = max(w, h)
                    sq = Image.new(image.mode, (side, side), self._bg())
                    sq.paste(image, (0, (w - h) // 2) if w > h else ((h - w) // 2, 0))
                    image = sq
            image = self.augmentations(image=image)
        return self.batch(to_uint8_hwc(image)[None])[0]


class DummyRGBProcessor:
    def __call__(self, image) -> torch.Tensor:
        return torch.zeros(1)
