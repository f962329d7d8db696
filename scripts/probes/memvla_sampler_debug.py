#!/usr/bin/env python
"""MemVLA real-size fixture, frame 0: the one-launch sampler against the block-by-block sampler and against the fp64 restatements (bf16-rounded
operands / exact) on the SAME (noise, z, perceptual tokens) with the model's own weights.   python scripts/probes/memvla_sampler_debug.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import test_memvla_gpu as TM  # noqa: E402
from tests import test_kernels_gpu as TK  # noqa: E402
from dexbotic_amd.engine import Fp32View  # noqa: E402

gd = os.path.join(ROOT, "tests", "golden")
g, x, m = TM._real(gd, "bfloat16", False)
m.eval()
head = m.model.action_head
net = head.net
cap = {}
orig = net.ddim_sample_fused


def spy(noise, z, diffusion, cfg, per_token=None):
    cap.update(noise=noise.clone(), z=z.clone(), per=per_token.clone(), diffusion=diffusion, cfg=cfg)
    return orig(noise, z, diffusion, cfg, per_token=per_token)


net.ddim_sample_fused = spy
norms = {"min": [-1.0] * 7, "max": [1.0] * 7}
T = TM.T
m.inference_action(T(x["infer_prompt"]), T(x["infer_frames"][0:1]), "True", {"cfg_scale": 1.5, "num_ddim_steps": 10, "action_norms": norms, "use_graph": False},
                   noise=T(x["infer_inits"][0]))
net.ddim_sample_fused = orig
noise, z, per, diff, cfg = cap["noise"], cap["z"], cap["per"], cap["diffusion"], cap["cfg"]
print("z |max|", float(z.abs().max()), "per |max|", float(per.abs().max()), "cfg", cfg)
with torch.no_grad():
    one = orig(noise, z, diff, cfg, per_token=per)
    mk = dict(z=z, per_kv=net.precompute_per_kv(per), cfg_scale=cfg)
    n2 = torch.cat([noise, noise], 0)
    blk = diff.ddim_sample_loop(net.forward_with_cfg, n2.shape, n2, clip_denoised=False, model_kwargs=mk, eta=0.0, device=noise.device)[:1]
    # fp64 restatements with the model's weights
    st = Fp32View(net.store)
    p, h = net.p, net.hidden_size
    ws = []
    for k in range(net.depth):
        b = f"{p}blocks.{k}."
        ws.append([st.w(b + n) for n in ("attn.qkv.weight", "attn.qkv.bias", "attn.proj.weight", "attn.proj.bias", "mlp.fc1.weight", "mlp.fc1.bias",
                                          "mlp.fc2.weight", "mlp.fc2.bias", "per_attn.in_proj_weight", "per_attn.in_proj_bias",
                                          "per_attn.out_proj.weight", "per_attn.out_proj.bias", "norm3.weight", "norm3.bias")])
    kv = net.precompute_per_kv(per, packed=True)
    tv, coef = net._sampler_tables(diff, noise.device)
    from dexbotic_amd import functional as Fn, kernels as K, _lib as L
    anchor = net._anchor()
    ze = Fn.LinearFn.apply(z.reshape(2, net.token_size).float().contiguous(), anchor, st, p + "z_embedder.linear.weight", p + "z_embedder.linear.bias", L.ACT_NONE, None)
    tf = K.timestep_embedding(tv, net._timestep_freqs(noise.device))
    te = Fn.MlpFn.apply(tf, anchor, st, p + "t_embedder.mlp.0.weight", p + "t_embedder.mlp.0.bias", p + "t_embedder.mlp.2.weight", p + "t_embedder.mlp.2.bias", L.ACT_SILU)
    args = (noise.float(), ze, te, st.w(p + "positional_embedding").reshape(17, h), st.w(p + "x_embedder.linear.weight"), st.w(p + "x_embedder.linear.bias"),
            st.w(p + "final_layer.linear.weight"), st.w(p + "final_layer.linear.bias"), coef, ws, 2, net.num_heads, cfg)
    want = TK._dit_sampler_restated(*args, True, kv=kv)
    exact = TK._dit_sampler_restated(*args, False, kv=kv)
sc = float(exact.abs().max())
d = lambda a, b: float((a.double() - b.double()).abs().max()) / sc
print(f"|sample| max {sc:.3f}")
print(f"one launch vs its restatement {d(one, want):.2e} | vs exact {d(one, exact):.2e} | block by block vs exact {d(blk, exact):.2e} | one vs blocks {d(one, blk):.2e} | restatement vs exact {d(want, exact):.2e}")
