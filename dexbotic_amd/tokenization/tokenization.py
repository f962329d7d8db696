"""Prompt -> token ids with image placeholders: dexbotic/tokenization/tokenization.py:10-31 (host logic; the ids it
yields drive the device-side splice plan, dexbotic_amd/splice.py)."""
from __future__ import annotations

import torch

from ..constants import IMAGE_TOKEN_INDEX


def tokenizer_image_token(prompt, tokenizer, image_token_index=IMAGE_TOKEN_INDEX, return_tensors=None):
    pieces = [tokenizer(text).input_ids for text in prompt.split("<image>")]
    ids = []
    skip = 0
    # a tokenizer that prepends BOS does so for every piece: keep the first one only
    if pieces and pieces[0] and pieces[0][0] == tokenizer.bos_token_id:
        skip = 1
        ids.append(pieces[0][0])
    for n, piece in enumerate(pieces):
        if n:
            ids.append(image_token_index)
        ids.extend(piece[skip:])
    if return_tensors is None:
        return ids
    if return_tensors == "pt":
        return torch.tensor(ids, dtype=torch.long)
    raise ValueError(f"Unsupported tensor type: {return_tensors}")
