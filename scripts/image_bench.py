#!/usr/bin/env python
"""Input-pipeline micro-benchmark (§8(f) rank 4): frames/s of the device preprocessing (uint8 frames already in HBM ->
normalised [n,3,224,224]) for one training micro-batch (16 episodes x 2 views of 480x640), HIP-event timed, next to the
host path the reference runs per frame (Pillow + CLIP image processor; 1 core) on the same frames.
    python scripts/image_bench.py [n_frames] [h] [w]"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dexbotic_amd.data.dataset.rgb_preprocess import ImageProcessorSpec, PreprocessRGB  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    h = int(sys.argv[2]) if len(sys.argv) > 2 else 480
    w = int(sys.argv[3]) if len(sys.argv) > 3 else 640
    rs = np.random.RandomState(0)
    frames = rs.randint(0, 256, (n, h, w, 3)).astype(np.uint8)
    pre = PreprocessRGB(ImageProcessorSpec(), image_aspect_ratio="pad", device="cuda")
    dev_frames = torch.from_numpy(frames).cuda()
    for _ in range(3):
        pre.batch(dev_frames)
    torch.cuda.synchronize()
    reps = 50
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = pre.batch(dev_frames)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    pinned = torch.from_numpy(frames).pin_memory()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        out = pre.batch(pinned)
    torch.cuda.synchronize()
    ms_h2d = 1e3 * (time.perf_counter() - t0) / 10
    algo = n * (h * w * 3 + 3 * 224 * 224 * 4)                      # frame bytes read + float32 planes written
    res = {"workload": f"{n} frames {h}x{w} -> pad -> 224x224 fp32", "device_ms": round(ms, 4),
           "frames_per_s": round(n / ms * 1e3, 1), "algorithmic_GBps": round(algo / ms / 1e6, 2),
           "with_pcie_upload_ms": round(ms_h2d, 4), "frames_per_s_with_upload": round(n / ms_h2d * 1e3, 1)}
    try:                                                            # host path of the reference, one core
        from PIL import Image
        from transformers import CLIPImageProcessor
        proc = CLIPImageProcessor()
        mean = tuple(int(x * 255) for x in proc.image_mean)
        k = min(n, 16)
        t0 = time.perf_counter()
        for i in range(k):
            im = Image.fromarray(frames[i])
            side = max(im.size)
            sq = Image.new("RGB", (side, side), mean)
            sq.paste(im, ((side - im.size[0]) // 2, (side - im.size[1]) // 2))
            ref = proc.preprocess(sq, return_tensors="pt")["pixel_values"][0]
        cpu_ms = 1e3 * (time.perf_counter() - t0) / k
        res["host_reference_ms_per_frame"] = round(cpu_ms, 3)
        res["host_reference_frames_per_s_1core"] = round(1e3 / cpu_ms, 1)
        res["max_abs_diff_vs_host"] = float((out[k - 1].cpu() - ref).abs().max())
    except Exception as e:                                          # pragma: no cover
        res["host_reference"] = f"unavailable: {e}"
    print(json.dumps(res))


if __name__ == "__main__":
    main()
