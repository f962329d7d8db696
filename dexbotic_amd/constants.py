"""Sentinel values of the token stream, as the reference's checkpoints and datasets use them (dexbotic/constants.py).

They are data-format constants, not code: the splice plan (dexbotic_amd/splice.py), the cross-entropy kernel
(dxa_cross_entropy_fwd's ignore_index) and the collator all have to agree with what the reference writes into
``input_ids`` / ``labels``.
"""

# labels equal to this take no part in the loss (torch.nn.CrossEntropyLoss's default ignore_index)
IGNORE_INDEX: int = -100

# placeholder id standing for ONE image in input_ids; the splice replaces it by that image's patch embeddings
IMAGE_TOKEN_INDEX: int = -200

# the text marker tokenizer_image_token() turns into IMAGE_TOKEN_INDEX
DEFAULT_IMAGE_TOKEN: str = "<image>"

__all__ = ["IGNORE_INDEX", "IMAGE_TOKEN_INDEX", "DEFAULT_IMAGE_TOKEN"]
