#!/usr/bin/env python
"""The CPU baseline of bench.py run on the REFERENCE'S OWN CLASSES (kind "reference"): dexbotic's CogACTForCausalLM +
clip_grad_norm_ + torch.optim.AdamW, imported from /root/reference — build container only (the GPU box has no reference tree
and times the port there).  Writes profiles/r03_cpu_reference_baseline.json.

    python scripts/cpu_reference_baseline.py [--port]       # --port: the CPU oracle on the same host, for the ratio"""
import json
import os
import platform
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    from dexbotic_amd.model.llm.qwen2 import Qwen2Config
    from dexbotic_amd.model.modules.mm_vision.clip.clip_encoder import CLIPVisionConfig
    port = "--port" in sys.argv
    args = types.SimpleNamespace(cpu_port=port, cpu_threads=0, views=1, s_text=32, batch=16)
    res = bench.cpu_baseline(args, Qwen2Config(), CLIPVisionConfig())
    res["host"] = {"nproc": os.cpu_count(), "machine": platform.processor() or platform.machine()}
    try:
        with open("/proc/cpuinfo") as f:
            res["host"]["model"] = next(l.split(":", 1)[1].strip() for l in f if l.startswith("model name"))
    except Exception:  # noqa: BLE001
        pass
    dst = os.path.join(ROOT, "profiles", "r03_cpu_port_baseline_container.json" if port else "r03_cpu_reference_baseline.json")
    with open(dst, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
