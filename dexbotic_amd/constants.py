"""Token constants shared with the reference (dexbotic/constants.py:1-3)."""
IGNORE_INDEX = -100
IMAGE_TOKEN_INDEX = -200
DEFAULT_IMAGE_TOKEN = "<image>"
