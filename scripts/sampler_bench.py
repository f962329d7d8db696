#!/usr/bin/env python
"""The DDIM sampler of a single request at DiT-B size (CFG batch 2 x 17 tokens, 10 steps): ONE persistent launch
(dxa_dit_sample_fwd) against the per-step path (persistent blocks + ~14 small launches per step).  GPU time by HIP events."""
import os
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dexbotic_amd.engine import ParamStore, attach_parameters, building  # noqa: E402
from dexbotic_amd.model.cogact.action_model.builder import build_action_model  # noqa: E402


def main():
    dev = "cuda"
    st = ParamStore(dev, torch.float32)
    with building(st):
        head = build_action_model(types.SimpleNamespace(action_model_type="DiT-B", hidden_size=3584, action_dim=7, chunk_size=16))
    st.finalize(train=False)
    root = torch.nn.Module()
    attach_parameters(root, st)
    st.master.normal_(0.0, 0.02, generator=torch.Generator(device=dev).manual_seed(0))
    head.eval()
    head.create_ddim(10)
    z = torch.randn(2, 1, 3584, device=dev)
    noise = torch.randn(1, 16, 7, device=dev)

    @torch.no_grad()
    def fused():
        return head.net.ddim_sample_fused(noise, z, head.ddim_diffusion, 1.5)

    @torch.no_grad()
    def per_step():
        n2 = torch.cat([noise, noise], 0)
        return head.ddim_diffusion.ddim_sample_loop(head.net.forward_with_cfg, n2.shape, n2, clip_denoised=False,
                                                    model_kwargs=dict(z=z, cfg_scale=1.5), eta=0.0, device=dev)[:1]
    a, b = fused(), per_step()
    print("max abs diff:", float((a - b).abs().max()), "scale", float(b.abs().max()))
    for name, fn in (("one launch", fused), ("per step", per_step)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(f"{name}: {e0.elapsed_time(e1) / 20:.3f} ms per 10-step sample", flush=True)


if __name__ == "__main__":
    main()
