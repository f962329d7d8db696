#!/usr/bin/env python
"""The one-launch DDIM sampler with bf16 MFMA operands (dxa_dit_sample_bf16_fwd) at DiT-B size: against a torch restatement of ITS
arithmetic (operands rounded to bf16, fp32 everything else: tight), against the exact-fp32 one-launch sampler (the distance the
rounding costs), and timed beside it."""
import os
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dexbotic_amd.engine import Fp32View, ParamStore, attach_parameters, building  # noqa: E402
from dexbotic_amd.model.cogact.action_model.builder import build_action_model  # noqa: E402


def rb(t):
    return t.to(torch.bfloat16).to(torch.float64)


@torch.no_grad()
def emulate(head, noise, z, cfg_scale, round_ops=True):
    """float64 restatement of dit_sample_bf16_k: DiT.forward_with_cfg (dit.py:273-311) inside ddim_sample_loop (diffusion.py:714-794)"""
    net, st = head.net, Fp32View(head.net.store)
    p, H, heads, depth = net.p, net.hidden_size, net.num_heads, net.depth
    W = lambda n: st.w(p + n).double()
    R = rb if round_ops else (lambda t: t.double())
    from dexbotic_amd import kernels as K
    from dexbotic_amd import functional as Fn, _lib as L
    tv, coef = net._sampler_tables(head.ddim_diffusion, noise.device)
    anchor = net._anchor()
    ze = Fn.LinearFn.apply(z.reshape(z.shape[0], -1).float().contiguous(), anchor, st, p + "z_embedder.linear.weight", p + "z_embedder.linear.bias", L.ACT_NONE, None).double()
    tf = K.timestep_embedding(tv, net._timestep_freqs(noise.device))
    te = Fn.MlpFn.apply(tf, anchor, st, p + "t_embedder.mlp.0.weight", p + "t_embedder.mlp.0.bias", p + "t_embedder.mlp.2.weight", p + "t_embedder.mlp.2.bias", L.ACT_SILU).double()
    x = noise.double().clone()
    nb, T, A = x.shape
    N = z.shape[0]
    pos = W("positional_embedding").reshape(T + 1, H)
    ln = lambda h: (h.mean(-1, keepdim=True), (h.var(-1, unbiased=False, keepdim=True) + 1e-6).rsqrt())
    for s in range(tv.shape[0]):
        xe = x @ W("x_embedder.linear.weight").T + W("x_embedder.linear.bias")             # [nb, T, H]
        xe = xe.repeat(N // nb, 1, 1)
        h = torch.cat([(te[s][None, :] + ze)[:, None, :], xe], 1) + pos[None]               # [N, T+1, H]
        for k in range(depth):
            b = f"blocks.{k}."
            mu, rs = ln(h)
            Wq = R(W(b + "attn.qkv.weight"))
            qkv = rs * (R(h) @ Wq.T - mu * Wq.sum(1)) + W(b + "attn.qkv.bias")
            q, kk, v = qkv.reshape(N, T + 1, 3, heads, 64).permute(2, 0, 3, 1, 4)
            att = torch.softmax((q @ kk.transpose(-1, -2)) * 0.125, -1) @ v                   # [N, heads, T+1, 64]
            o = att.permute(0, 2, 1, 3).reshape(N, T + 1, H)
            h = h + R(o) @ R(W(b + "attn.proj.weight")).T + W(b + "attn.proj.bias")
            mu, rs = ln(h)
            W1 = R(W(b + "mlp.fc1.weight"))
            a = torch.nn.functional.gelu(rs * (R(h) @ W1.T - mu * W1.sum(1)) + W(b + "mlp.fc1.bias"), approximate="tanh")
            h = h + R(a) @ R(W(b + "mlp.fc2.weight")).T + W(b + "mlp.fc2.bias")
        mu, rs = ln(h)
        eps = ((h - mu) * rs) @ W("final_layer.linear.weight").T + W("final_layer.linear.bias")
        eps = eps[:, 1:, :]
        if cfg_scale is not None:
            c, u = eps[:nb], eps[nb:]
            eps = u + cfg_scale * (c - u)
        c0, c1, ab = (coef[s, i].double() for i in range(3))
        x0 = c0 * x - c1 * eps
        e2 = (c0 * x - x0) / c1
        x = x0 * ab.sqrt() + (1 - ab).sqrt() * e2
    return x.float()


def main():
    dev = "cuda"
    st = ParamStore(dev, torch.bfloat16)
    with building(st):
        head = build_action_model(types.SimpleNamespace(action_model_type=os.environ.get("DIT", "DiT-B"), hidden_size=3584, action_dim=7, chunk_size=16))
    st.finalize(train=False)
    root = torch.nn.Module()
    attach_parameters(root, st)
    st.master.normal_(0.0, 0.02, generator=torch.Generator(device=dev).manual_seed(0))
    head.eval()
    head.create_ddim(10)
    z = torch.randn(2, 1, 3584, device=dev)
    noise = torch.randn(1, 16, 7, device=dev)
    rel = lambda a, b: float((a.double() - b.double()).abs().max() / b.double().abs().max())

    @torch.no_grad()
    def fused():
        return head.net.ddim_sample_fused(noise, z, head.ddim_diffusion, 1.5)
    assert head.net._bf16_sampler(2, 17)
    a = fused()
    a2 = fused()
    os.environ["DXA_DIT_BF16"] = "0"
    b = fused()
    os.environ["DXA_DIT_BF16"] = "1"
    e = emulate(head, noise, z, 1.5)
    e32 = emulate(head, noise, z, 1.5, round_ops=False)
    print(f"bf16-operand sampler: run-to-run identical {bool(torch.equal(a, a2))}; vs torch restatement of its arithmetic {rel(a, e):.2e}; "
          f"vs exact-fp32 sampler {rel(a, b):.2e}; exact-fp32 sampler vs fp64 torch {rel(b, e32):.2e}; restatement bf16 vs fp64 {rel(e, e32):.2e}")
    for name, env in (("bf16 operands", "1"), ("fp32 exact", "0")):
        os.environ["DXA_DIT_BF16"] = env
        for _ in range(5):
            fused()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fused()
        e1.record()
        torch.cuda.synchronize()
        print(f"{name}: {e0.elapsed_time(e1) / 20:.3f} ms per 10-step sample", flush=True)


if __name__ == "__main__":
    main()
