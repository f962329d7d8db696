"""pi0 golden vectors at the REAL widths from the reference's own Pi0ForCausalLM — TEST INFRASTRUCTURE.

    python -m oracle.gen_golden_pi0_real       # build container only (needs /root/reference) -> tests/golden/pi0_real_ref.npz

PaliGemma-3B / pi0 widths (pi0_arch.py:58-110 defaults): Gemma-2B expert d 2048, ffn 16384, 8 q / 1 kv heads x 256; action
expert d 1024, ffn 4096; SigLIP-So400m tower d 1152, ffn 4304, 16 heads x 72, 224 px / patch 14 (256 tokens per camera);
action_dim 32, the reference's default chunk_size 50.  Shallow: 2 mixture layers, 2 tower layers, 2048-row vocabulary.
B = 2, 3 cameras (one masked), 16-token instruction with one right-padded sample => prefix 784, suffix 51.
Three runs:
  * "fp32" training step (loss, v_t, per-group gradient norms, strided gradient samples);
  * "bf16" training step under torch.autocast("cpu", bfloat16) (HF Trainer bf16=True, fp32 weights);
  * fp32 inference_action (the reference samples in fp32, pi0_exp.py:347-353), 10 Euler steps, chunk 50.
The hd-256 block-masked flash kernels and the SigLIP hd-72-on-128 padding are exactly what bench.py's secondary pi0 line
runs; tests/golden/pi0_t1.npz (tiny) reaches neither."""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

from .gen_golden_pi0 import GOLD, REF, ROOT, build_reference, inject
from .pi0_oracle import Pi0OracleConfig, pi0_shapes
from .weights import make_weights, weights_crc

REAL = Pi0OracleConfig(vocab_size=2048, hidden_size=2048, intermediate_size=16384, num_hidden_layers=2,
                       num_attention_heads=8, num_key_value_heads=1, head_dim=256, a_hidden=1024, a_inter=4096,
                       v_hidden=1152, v_inter=4304, v_layers=2, v_heads=16, v_image=224, v_patch=14, action_dim=32,
                       chunk_size=50)
SEED = 4242
STRIDE = 499
GROUPS = {"llm": "model.llm.", "expert": "model.action_expert.", "vision": "model.mm_vision_tower.",
          "rest": ("model.mm_projector.", "model.state_proj.", "model.action_in_proj.", "model.action_out_proj.",
                   "model.action_time_mlp_in.", "model.action_time_mlp_out.")}
GSAMP = ("model.llm.layers.0.self_attn.q_proj.weight", "model.llm.layers.0.mlp.down_proj.weight",
         "model.llm.layers.1.self_attn.k_proj.weight", "model.action_expert.layers.1.mlp.gate_proj.weight",
         "model.action_expert.layers.0.self_attn.o_proj.weight", "model.action_out_proj.weight",
         "model.mm_vision_tower.vision_tower.encoder.layers.0.self_attn.q_proj.weight",
         "model.mm_vision_tower.vision_tower.encoder.layers.1.mlp.fc2.weight", "model.mm_projector.weight")


def inputs(cfg=REAL):
    rs = np.random.RandomState(11)
    B, L, CAM = 2, 16, 3
    ids = rs.randint(5, cfg.vocab_size - 5, size=(B, L)).astype(np.int64)
    mask = np.ones((B, L), dtype=bool)
    mask[1, 11:] = False
    image_masks = np.ones((B, CAM), dtype=bool)
    image_masks[0, 2] = False
    images = np.clip(rs.standard_normal((B, CAM, 3, cfg.v_image, cfg.v_image)), -2.5, 2.5).astype(np.float32)
    states = rs.standard_normal((B, cfg.action_dim)).astype(np.float32)
    actions = rs.standard_normal((B, cfg.chunk_size, cfg.action_dim)).astype(np.float32)
    noise = rs.standard_normal((B, cfg.chunk_size, cfg.action_dim)).astype(np.float32)
    tm = rs.uniform(0.05, 0.95, size=(B,)).astype(np.float32)
    init = rs.standard_normal((B, cfg.chunk_size, cfg.action_dim)).astype(np.float32)
    return dict(input_ids=ids, attention_mask=mask, image_masks=image_masks, images=images, states=states, actions=actions,
                noise=noise, time=tm, init_noise=init)


def summarize(named_grads, loss, v_t):
    res = {"loss": np.float64(loss), "v_t": np.asarray(v_t, dtype=np.float32)}
    for g, pre in GROUPS.items():
        sq = sum(float(v.double().pow(2).sum()) for n, v in named_grads.items() if v is not None and n.startswith(pre))
        res[f"gnorm/{g}"] = np.float64(sq ** 0.5)
    for n in GSAMP:
        res["gsamp/" + n] = named_grads[n].reshape(-1)[::STRIDE].float().cpu().numpy().copy()
    return res


def main():
    sys.path.insert(0, REF)
    sys.path.insert(0, ROOT)
    torch.set_num_threads(os.cpu_count() or 8)
    from .gen_golden import install_timm_shim
    install_timm_shim()
    t0 = time.time()
    w = make_weights(pi0_shapes(REAL), SEED)
    m = build_reference(REAL, w)
    x = inputs()
    t = torch.from_numpy
    import zlib
    res = {"seed": np.int64(SEED), "weights_crc": np.int64(weights_crc(w)),
           "images_crc": np.int64(zlib.crc32(x["images"].tobytes()))}
    res.update({k: v for k, v in x.items() if k != "images"})
    for tag, ac in (("fp32", False), ("bf16", True)):
        m.train()
        m.zero_grad(set_to_none=True)
        with inject(t(x["noise"]), t(x["time"])), torch.autocast("cpu", dtype=torch.bfloat16, enabled=ac):
            out = m(input_ids=t(x["input_ids"]), attention_mask=t(x["attention_mask"]), images=t(x["images"]),
                    image_masks=t(x["image_masks"]), states=t(x["states"]), actions=t(x["actions"]))
        out.loss.backward()
        r = summarize({n: p.grad for n, p in m.named_parameters()}, out.loss.item(), out.logits.detach().float().numpy())
        print(tag, "loss", float(r["loss"]), {k: round(float(v), 5) for k, v in r.items() if k.startswith("gnorm/")},
              f"{time.time()-t0:.0f}s", flush=True)
        for k, v in r.items():
            res[f"{tag}/{k}"] = v
    m.zero_grad(set_to_none=True)
    m.eval()
    with torch.no_grad(), inject(t(x["init_noise"])):
        acts = m.inference_action(input_ids=t(x["input_ids"]), attention_mask=t(x["attention_mask"]), states=t(x["states"]),
                                  images=t(x["images"]), image_masks=t(x["image_masks"]), diffusion_steps=10)
    res["fp32/infer_actions"] = acts.numpy().astype(np.float32)
    print("infer |a|", float(np.abs(res["fp32/infer_actions"]).mean()), f"{time.time()-t0:.0f}s")
    # the CPU oracle against the same vectors, for the record (tests/test_pi0_oracle.py recomputes the fp32 part)
    from . import pi0_oracle as P
    sd = {k: torch.from_numpy(v).requires_grad_(True) for k, v in w.items()}
    o = P.pi0_forward(sd, REAL, t(x["input_ids"]), t(x["attention_mask"]), t(x["images"]), t(x["image_masks"]),
                      t(x["states"]), t(x["actions"]), t(x["noise"]), t(x["time"]))
    o["loss"].backward()
    ro = summarize({n: p.grad for n, p in sd.items()}, o["loss"].item(), o["v_t"].detach().numpy())
    rel = lambda a, b: float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() /
                             (np.abs(np.asarray(b, np.float64)).max() + 1e-12))
    d = {k: rel(v, res["fp32/" + k]) for k, v in ro.items()}
    print("oracle vs reference fp32", {k: f"{v:.1e}" for k, v in d.items()})
    for k, v in d.items():
        res["oracle_vs_ref/fp32/" + k] = np.float64(v)
    dst = os.path.join(GOLD, "pi0_real_ref.npz")
    np.savez_compressed(dst, **res)
    print("wrote", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
