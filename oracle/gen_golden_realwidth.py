"""Golden vectors at the BASELINE widths (d 3584, 28 q / 4 kv heads x 128, ffn 18944, CLIP-L 1024/16/4096 @224,
DiT-B 768 x 12) with a shallow stack (2 decoder layers, 2 used + 1 unused ViT layers) and a small vocabulary (the embedding
is a row gather: its height is not a kernel width):

    python -m oracle.gen_golden_realwidth          # ~2-4 min on 8 cores -> tests/golden/cogact_real.npz (a few KB)

The tiny-shape fixtures (cogact_t1/t2) never reach the 256-row MFMA tiles, the split-K tails, the 7:1 GQA flash kernels or
the bf16x3 head at their real shapes; this one does, end to end.  Two references per quantity:
  * fp32: oracle.cogact_forward / backward in float32 (bar: 1e-3 relative, the north-star tolerance);
  * bf16: the same oracle under torch.autocast(bfloat16) for tower + projector + decoder — how the reference trains
    (trainer bf16=True; the action head stays fp32 under autocast(float32), cogact_arch.py:133) — the yardstick for the
    product's bf16 compute mode instead of an fp32 golden at a loose bound.
Stored: losses, cognition features, eps_hat, gradient norms per module group + strided gradient samples, the DDIM result of
a 2-view B=1 inference.  Weights are NOT stored (make_weights(seed) regenerates them bit-exactly; weights_crc pins that).
TEST INFRASTRUCTURE: the oracle is the checker, never the product."""
import os
import time

import numpy as np
import torch

from . import cogact_oracle as O
from .weights import cogact_shapes, make_weights, weights_crc

REAL = O.OracleConfig(vocab_size=2048, hidden_size=3584, intermediate_size=18944, num_hidden_layers=2,
                      num_attention_heads=28, num_key_value_heads=4, v_hidden=1024, v_inter=4096, v_layers=3, v_heads=16,
                      v_image=224, v_patch=14, dit_hidden=768, dit_depth=12, dit_heads=12)
SEED = 21
GROUPS = {"llm": "model.llm.", "vision": "model.mm_vision_tower.", "projector": "model.mm_projector.",
          "head": "model.action_head."}


def inputs():
    rs = np.random.RandomState(5)
    B, St = 2, 32
    ids = rs.randint(10, REAL.vocab_size, size=(B, St)).astype(np.int64)
    ids[:, 1] = -200
    mask = np.ones((B, St), dtype=bool)
    mask[1, 27:] = False                                   # one right-padded sample
    images = np.clip(rs.standard_normal((B, 3, 224, 224)), -2.5, 2.5).astype(np.float32)
    actions = rs.uniform(-1, 1, size=(B, 112)).astype(np.float32)
    noise = rs.standard_normal((4 * B, 16, 7)).astype(np.float32)
    ts = rs.randint(0, 100, size=(4 * B,)).astype(np.int64)
    drop_u = rs.uniform(size=(4 * B,)).astype(np.float32)
    images2 = np.clip(rs.standard_normal((1, 2, 3, 224, 224)), -2.5, 2.5).astype(np.float32)   # B = 1, 2 views
    init = rs.standard_normal((1, 16, 7)).astype(np.float32)
    return dict(input_ids=ids, attention_mask=mask, images=images, actions=actions, noise=noise, timesteps=ts,
                drop_u=drop_u, infer_ids=ids[:1].copy(), infer_images=images2, infer_init=init)


def run(sd, x, autocast: bool):
    t = torch.from_numpy
    for p in sd.values():
        p.grad = None
    with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
        feats = O.extract_vision_features(sd, REAL, t(x["images"]))
        src, new_mask, _ = O.splice_plan(x["input_ids"], x["attention_mask"], feats.shape[1], None, "right")
        # torch.cat of fp32 embedding rows and bf16 image features promotes to fp32 (dexbotic_arch.py:291-304)
        hidden = O.qwen2_forward(sd, REAL, O.splice_embeds(sd, src, feats.float()), t(new_mask))
        cog = O.cognition_features(hidden, t(new_mask))
    with torch.autocast("cpu", enabled=False):
        loss, x_t, eps_hat = O.action_loss(sd, REAL, t(x["actions"]), cog.float(), t(x["noise"]), t(x["timesteps"]),
                                           t(x["drop_u"]) < 0.1, 4)
    loss.backward()
    out = {"loss": loss.item(), "cognition": cog.detach().float().numpy(), "eps_hat": eps_hat.detach().numpy()}
    for g, pre in GROUPS.items():
        sq = sum(float(p.grad.double().pow(2).sum()) for n, p in sd.items() if n.startswith(pre) and p.grad is not None)
        out[f"gnorm/{g}"] = sq ** 0.5
    for n in ("model.llm.layers.0.self_attn.q_proj.weight", "model.llm.layers.1.mlp.down_proj.weight",
              "model.llm.layers.0.self_attn.k_proj.bias", "model.llm.layers.1.input_layernorm.weight",
              "model.mm_projector.2.weight", "model.mm_vision_tower.vision_tower.encoder.layers.0.mlp.fc1.weight",
              "model.action_head.net.blocks.11.mlp.fc2.weight", "model.action_head.net.z_embedder.linear.weight"):
        out["gsamp/" + n] = sd[n].grad.reshape(-1)[::997].float().numpy().copy()
    with torch.no_grad():
        with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
            io = O.cogact_forward(sd, REAL, t(x["infer_ids"]), None, t(x["infer_images"]))
            cog1 = io["logits"][:, -1, :][:, None, :].float()
        with torch.autocast("cpu", enabled=False):
            samples = O.ddim_sample(sd, REAL, cog1, t(x["infer_init"]), 1.5, 10)
    out["infer_cognition"] = cog1.numpy()
    out["infer_samples"] = samples.numpy()
    return out


def main():
    torch.set_num_threads(os.cpu_count() or 8)
    t0 = time.time()
    w = make_weights(cogact_shapes(REAL), SEED)
    sd = {k: torch.from_numpy(v).requires_grad_(True) for k, v in w.items()}
    x = inputs()
    res = {"seed": np.int64(SEED), "weights_crc": np.int64(weights_crc(w))}
    res.update(x)
    for tag, ac in (("fp32", False), ("bf16", True)):
        r = run(sd, x, ac)
        print(tag, "loss", r["loss"], {k: round(v, 5) for k, v in r.items() if k.startswith("gnorm/")}, f"{time.time()-t0:.0f}s",
              flush=True)
        for k, v in r.items():
            res[f"{tag}/{k}"] = np.asarray(v)
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "cogact_real.npz")
    np.savez_compressed(dst, **res)
    print("wrote", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
