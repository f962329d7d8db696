#!/usr/bin/env python
"""dxa_transpose micro-benchmark on the DB-CogACT step's shapes (bf16): achieved read+write TB/s."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dexbotic_amd import kernels as K  # noqa: E402

SHAPES = [("qkv W", 4608, 3584), ("o W", 3584, 3584), ("gate_up W", 37888, 3584), ("down W", 3584, 18944),
          ("X d", 4592, 3584), ("X ffn", 4592, 18944), ("dY qkv", 4592, 4608), ("dY gate_up", 4592, 37888),
          ("vit X", 4112, 1024), ("vit ffn", 4112, 4096)]


def main():
    for name, R, C in SHAPES:
        x = torch.randn(R, C, device="cuda").bfloat16()
        for _ in range(3):
            y = K.transpose(x, 64)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            y = K.transpose(x, 64)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"{name:12s} [{R:6d},{C:6d}] {ms * 1e3:8.1f} us  {(x.numel() + y.numel()) * 2 / ms / 1e9:6.2f} TB/s", flush=True)


if __name__ == "__main__":
    main()
