"""Golden vectors at the BASELINE widths from the REFERENCE'S OWN CLASSES (not from the oracle):

    python -m oracle.gen_golden_realwidth_ref      # ~6-10 min on 8 cores -> tests/golden/cogact_real_ref.npz

dexbotic.model.cogact.cogact_arch.CogACTForCausalLM (imported from /root/reference with the shims of oracle/gen_golden.py) at
d 3584, 28 q / 4 kv heads x 128, ffn 18944, CLIP-L 1024/16/4096 @224, DiT-B 768 x 12, FOUR decoder layers, 2 used + 1 unused
ViT layers, a 2048-row vocabulary (a row gather: its height is not a kernel width), B = 2 with one right-padded sample,
S = 287.  Two runs of the same step:
  * "fp32": plain float32 (bar for the product's fp32 mode and for the oracle: 1e-3 relative);
  * "bf16": ``with torch.autocast("cpu", dtype=torch.bfloat16)`` around ``model(**inputs)`` with fp32 weights — what HF
    ``Trainer.compute_loss`` does for ``bf16=True`` (DexboticTrainer, exp/trainer.py:100-124; HF wraps the forward in
    ``accelerator.autocast()``), backward outside autocast.  This is the yardstick the product's bf16 mode (the mode bench.py
    times) and the oracle under autocast are BOTH held to.
One documented shim for the bf16 run: the reference opens ``torch.amp.autocast('cuda', dtype=torch.float32)`` around the
action-head loss (cogact_arch.py:133), which on a GPU turns autocast off for the head (fp32 head).  On the CPU that context
manager does not touch the "cpu" autocast state, so the head would silently run in bf16 here; the generator therefore runs
``action_head.loss`` (and the DDIM loop of inference) under ``torch.autocast("cpu", enabled=False)`` with fp32 inputs — the
arithmetic the reference performs on its real device.
Stored: loss, cognition features, eps_hat, per-group gradient norms, strided gradient samples, the result of a 2-view B = 1
``inference_action`` (CFG 1.5, 10 DDIM steps, injected initial noise).  Weights are regenerated from the seed
(oracle/weights.make_weights; weights_crc pins that).  The distances of the ORACLE to these vectors are printed and stored
under "oracle_vs_ref/*" for the record; tests/test_oracle_realwidth.py recomputes them.
TEST INFRASTRUCTURE: runs only in the build container (needs /root/reference)."""
import os
import sys
import time
import zlib

import numpy as np
import torch

from . import cogact_oracle as O
from . import gen_golden as G
from .weights import cogact_shapes, make_weights, weights_crc

REAL4 = O.OracleConfig(vocab_size=2048, hidden_size=3584, intermediate_size=18944, num_hidden_layers=4,
                       num_attention_heads=28, num_key_value_heads=4, v_hidden=1024, v_inter=4096, v_layers=3, v_heads=16,
                       v_image=224, v_patch=14, dit_hidden=768, dit_depth=12, dit_heads=12)
SEED = 23
GROUPS = {"llm": "model.llm.", "vision": "model.mm_vision_tower.", "projector": "model.mm_projector.",
          "head": "model.action_head."}
GSAMP = ("model.llm.layers.0.self_attn.q_proj.weight", "model.llm.layers.3.mlp.down_proj.weight",
         "model.llm.layers.2.mlp.gate_proj.weight", "model.llm.layers.1.self_attn.o_proj.weight",
         "model.llm.layers.0.self_attn.k_proj.bias", "model.llm.layers.3.input_layernorm.weight",
         "model.mm_projector.2.weight", "model.mm_vision_tower.vision_tower.encoder.layers.0.mlp.fc1.weight",
         "model.action_head.net.blocks.11.mlp.fc2.weight", "model.action_head.net.z_embedder.linear.weight")
STRIDE = 997


def inputs():
    rs = np.random.RandomState(7)
    B, St = 2, 32
    ids = rs.randint(10, REAL4.vocab_size, size=(B, St)).astype(np.int64)
    ids[:, 1] = -200
    mask = np.ones((B, St), dtype=bool)
    mask[1, 25:] = False                                   # one right-padded sample
    images = np.clip(rs.standard_normal((B, 3, 224, 224)), -2.5, 2.5).astype(np.float32)
    actions = rs.uniform(-1, 1, size=(B, 112)).astype(np.float32)
    noise = rs.standard_normal((4 * B, 16, 7)).astype(np.float32)
    ts = rs.randint(0, 100, size=(4 * B,)).astype(np.int64)
    drop_u = rs.uniform(size=(4 * B,)).astype(np.float32)
    drop_u[3] = 0.01                                       # at least one dropped condition
    images2 = np.clip(rs.standard_normal((1, 2, 3, 224, 224)), -2.5, 2.5).astype(np.float32)   # B = 1, 2 views
    init = rs.standard_normal((1, 16, 7)).astype(np.float32)
    return dict(input_ids=ids, attention_mask=mask, images=images, actions=actions, noise=noise, timesteps=ts,
                drop_u=drop_u, infer_ids=ids[:1].copy(), infer_images=images2, infer_init=init)


class fp32_head:
    """the documented shim (module docstring): run the action head outside the CPU autocast, as the reference's
    autocast('cuda', float32) does on its real device"""

    def __init__(self, m):
        self.head = m.model.action_head
        self.m = m

    def __enter__(self):
        self._loss = self.head.loss

        def loss(x, z, *a, **k):
            with torch.autocast("cpu", enabled=False):
                return self._loss(x.float(), z.float(), *a, **k)
        self.head.loss = loss
        return self

    def __exit__(self, *exc):
        self.head.loss = self._loss


def run_reference(m, x, autocast: bool):
    t = torch.from_numpy
    m.train()
    m.zero_grad(set_to_none=True)
    cap = {}
    h = m.model.action_head.net.register_forward_hook(lambda mod, i, o: cap.__setitem__("eps_hat", o.detach().float()))
    with G.inject_rng(noise=t(x["noise"]), timesteps=t(x["timesteps"]), drop_u=t(x["drop_u"])), fp32_head(m):
        with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
            out = m(input_ids=t(x["input_ids"]), attention_mask=t(x["attention_mask"]), images=t(x["images"]),
                    actions=t(x["actions"]), labels=t(x["input_ids"]).clone())
    h.remove()
    out.loss.backward()
    hid = out.logits.detach().float()
    # the spliced mask: text length - 1 + N_v valid tokens per sample, right padding (dexbotic_arch.py:331-373)
    nv = REAL4.num_patches
    lens = x["attention_mask"].sum(1) - 1 + nv
    cog = torch.stack([hid[b, int(lens[b]) - 1] for b in range(hid.shape[0])])[:, None, :]
    res = {"loss": out.loss.item(), "cognition": cog.numpy(), "eps_hat": cap["eps_hat"].numpy()}
    grads = {n: p.grad for n, p in m.named_parameters()}
    for g, pre in GROUPS.items():
        sq = sum(float(v.double().pow(2).sum()) for n, v in grads.items() if n.startswith(pre) and v is not None)
        res[f"gnorm/{g}"] = sq ** 0.5
    for n in GSAMP:
        res["gsamp/" + n] = grads[n].reshape(-1)[::STRIDE].float().numpy().copy()
    m.zero_grad(set_to_none=True)
    # inference: 2 views, B = 1 (BASELINE.json configs[1])
    m.eval()
    head = m.model.action_head
    if head.ddim_diffusion is None:
        head.create_ddim(ddim_step=10)
    dd = head.ddim_diffusion
    orig_loop = dd.ddim_sample_loop
    got = {}

    def loop(fn, shape, noise, **k):
        with torch.autocast("cpu", enabled=False):
            mk = dict(k.pop("model_kwargs"))
            mk["z"] = mk["z"].float()
            got["z"] = mk["z"][:1].clone()
            s = orig_loop(fn, shape, noise.float(), model_kwargs=mk, **k)
        got["samples"] = s.detach().float().clone()
        return s
    dd.ddim_sample_loop = loop
    try:
        with G.inject_rng(init_noise=t(x["infer_init"])), torch.no_grad():
            with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
                m.inference_action(t(x["infer_ids"]), t(x["infer_images"]),
                                   {"cfg_scale": 1.5, "num_ddim_steps": 10,
                                    "action_norms": {"min": [-1.0] * 7, "max": [1.0] * 7}})
    finally:
        dd.ddim_sample_loop = orig_loop
    res["infer_cognition"] = got["z"].numpy()
    res["infer_samples"] = got["samples"][:1].numpy()
    return res


def run_oracle(sd, x, autocast: bool, cfg=REAL4):
    """the CPU restatement on the same inputs (same structure as oracle/gen_golden_realwidth.run)"""
    t = torch.from_numpy
    for p in sd.values():
        p.grad = None
    with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
        feats = O.extract_vision_features(sd, cfg, t(x["images"]))
        src, new_mask, _ = O.splice_plan(x["input_ids"], x["attention_mask"], feats.shape[1], None, "right")
        hidden = O.qwen2_forward(sd, cfg, O.splice_embeds(sd, src, feats.float()), t(new_mask))
        cog = O.cognition_features(hidden, t(new_mask))
    with torch.autocast("cpu", enabled=False):
        loss, x_t, eps_hat = O.action_loss(sd, cfg, t(x["actions"]), cog.float(), t(x["noise"]), t(x["timesteps"]),
                                           t(x["drop_u"]) < 0.1, 4)
    loss.backward()
    out = {"loss": loss.item(), "cognition": cog.detach().float().numpy(), "eps_hat": eps_hat.detach().numpy()}
    for g, pre in GROUPS.items():
        sq = sum(float(p.grad.double().pow(2).sum()) for n, p in sd.items() if n.startswith(pre) and p.grad is not None)
        out[f"gnorm/{g}"] = sq ** 0.5
    for n in GSAMP:
        out["gsamp/" + n] = sd[n].grad.reshape(-1)[::STRIDE].float().numpy().copy()
    with torch.no_grad():
        with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
            io = O.cogact_forward(sd, cfg, t(x["infer_ids"]), None, t(x["infer_images"]))
            cog1 = io["logits"][:, -1, :][:, None, :].float()
        with torch.autocast("cpu", enabled=False):
            samples = O.ddim_sample(sd, cfg, cog1, t(x["infer_init"]), 1.5, 10)
    out["infer_cognition"] = cog1.numpy()
    out["infer_samples"] = samples.numpy()
    return out


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


def distances(got, ref):
    return {k: rel(got[k], ref[k]) for k in ref}


def main():
    torch.set_num_threads(os.cpu_count() or 8)
    sys.path.insert(0, G.REF)
    G.install_timm_shim()
    t0 = time.time()
    w = make_weights(cogact_shapes(REAL4), SEED)
    m = G.build_reference(REAL4, w)
    x = inputs()
    res = {"seed": np.int64(SEED), "weights_crc": np.int64(weights_crc(w))}
    # the two image tensors (2.4 MB) are not stored: inputs() regenerates them (RandomState(7)); their CRCs pin that
    res.update({k: v for k, v in x.items() if k not in ("images", "infer_images")})
    res["images_crc"] = np.int64(zlib.crc32(x["images"].tobytes()))
    res["infer_images_crc"] = np.int64(zlib.crc32(x["infer_images"].tobytes()))
    refs = {}
    for tag, ac in (("fp32", False), ("bf16", True)):
        r = refs[tag] = run_reference(m, x, ac)
        print("reference", tag, "loss", r["loss"], {k: round(v, 5) for k, v in r.items() if k.startswith("gnorm/")},
              f"{time.time()-t0:.0f}s", flush=True)
        for k, v in r.items():
            res[f"{tag}/{k}"] = np.asarray(v)
    del m
    sd = {k: torch.from_numpy(v).requires_grad_(True) for k, v in w.items()}
    for tag, ac in (("fp32", False), ("bf16", True)):
        d = distances(run_oracle(sd, x, ac), refs[tag])
        print("oracle vs reference", tag, {k: f"{v:.2e}" for k, v in d.items()}, f"{time.time()-t0:.0f}s", flush=True)
        for k, v in d.items():
            res[f"oracle_vs_ref/{tag}/{k}"] = np.float64(v)
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "cogact_real_ref.npz")
    np.savez_compressed(dst, **res)
    print("wrote", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
