#!/usr/bin/env python
"""Single-GPU action-inference latency of the full-size DB-CogACT (batch 1, CFG 1.5, 10 DDIM steps):
p50 over 20 requests, eager launches vs HIP-graph replay.   python scripts/infer_bench.py [eager|graph|both]"""
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "both"
    dev = torch.device("cuda", 0)
    args = types.SimpleNamespace(llm_layers=28, vit_layers=24, dtype="bfloat16")
    model, cfg, _, _ = bench.build_model(args, dev)
    model.eval()
    views = int(os.environ.get("VIEWS", "2"))                  # BASELINE.json configs[1]: 2 views
    b1 = bench.synthetic_batch(1, views, 32, dev, seed=7)
    norms = {"min": [-1.0] * 7, "max": [1.0] * 7}
    for use_graph in ([False, True] if mode == "both" else [mode == "graph"]):
        lat = []
        for _ in range(int(os.environ.get("REQS", "25"))):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            model.inference_action(b1["input_ids"], b1["images"], {"cfg_scale": 1.5, "num_ddim_steps": 10,
                                                                    "action_norms": norms, "use_graph": use_graph})
            lat.append(1e3 * (time.perf_counter() - t0))
        print(f"{'graph' if use_graph else 'eager'}: p50 {np.median(lat[5:]):.2f} ms  min {min(lat[5:]):.2f} ms", flush=True)


if __name__ == "__main__":
    main()
