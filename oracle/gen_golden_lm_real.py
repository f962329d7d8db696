"""Golden vectors for the lm_head / cross-entropy / greedy-decode path (BASELINE.json configs[0]) at the
BASELINE widths, from the REFERENCE'S OWN CLASS:

    python -m oracle.gen_golden_lm_real      # ~5 min on 8 cores -> tests/golden/lm_real_ref.npz

dexbotic.model.dexbotic_arch.DexboticForCausalLM (imported from /root/reference with the shims of oracle/gen_golden.py) at
d 3584, 28 q / 4 kv heads x 128, ffn 18944, CLIP-L 1024/16/4096 @224 (2 used + 1 unused layers), FOUR decoder layers, a
4096-row vocabulary (lm_head [4096, 3584]: the vocabulary height is a row count, not a kernel width), B = 2 with one
right-padded sample, S = 287.  Stored:
  * "fp32/*": loss, logits (every 5th position + the last valid one), per-group gradient norms, strided gradient samples of a
    training step with labels (HF ForCausalLMLoss, dexbotic_arch.py:429-496) in plain float32 — the 1e-3 bar of the product's
    fp32 mode;
  * "bf16/*": the same step under ``torch.autocast("cpu", dtype=torch.bfloat16)`` (what HF Trainer does for bf16=True) — the
    yardstick for the product's bf16 mode, with "ref_bf16_vs_fp32/*" = the reference's own distance between the two;
  * a greedy continuation of sample 0 (batch 1, 4 new tokens) by full-prefix recompute through the reference forward in fp32
    (generate() itself does not run under this container's transformers, SURVEY.md §8c shim iii): ids, the logit rows and the
    top-1 / top-2 margins — a KV-cached decode must reproduce the ids exactly in fp32, and in bf16 wherever the margin exceeds the
    bf16 logit distance.
Weights are regenerated from the seed (oracle/weights.make_weights; weights_crc pins that).
TEST INFRASTRUCTURE: runs only in the build container (needs /root/reference)."""
import os
import sys
import tempfile
import time

import numpy as np
import torch

from . import cogact_oracle as O
from . import gen_golden as G
from .weights import cogact_shapes, make_weights, weights_crc

REAL4_LM = O.OracleConfig(vocab_size=4096, hidden_size=3584, intermediate_size=18944, num_hidden_layers=4,
                          num_attention_heads=28, num_key_value_heads=4, v_hidden=1024, v_inter=4096, v_layers=3, v_heads=16,
                          v_image=224, v_patch=14)
SEED = 29
GROUPS = {"llm": "model.llm.", "vision": "model.mm_vision_tower.", "projector": "model.mm_projector.", "lm_head": "lm_head."}
GSAMP = ("lm_head.weight", "model.llm.layers.0.self_attn.q_proj.weight", "model.llm.layers.3.mlp.down_proj.weight",
         "model.llm.layers.2.mlp.gate_proj.weight", "model.llm.layers.1.self_attn.o_proj.weight",
         "model.llm.layers.0.self_attn.k_proj.bias", "model.llm.norm.weight", "model.mm_projector.2.weight")
STRIDE = 997
POS_STRIDE = 5
N_NEW = 4


def lm_weights():
    return {k: v for k, v in make_weights(cogact_shapes(REAL4_LM), SEED).items() if ".action_head." not in k}


def inputs():
    rs = np.random.RandomState(11)
    B, St = 2, 32
    ids = rs.randint(10, REAL4_LM.vocab_size - 10, size=(B, St)).astype(np.int64)
    ids[:, 1] = -200
    mask = np.ones((B, St), dtype=bool)
    mask[1, 26:] = False                                   # one right-padded sample
    labels = ids.copy()
    labels[:, :4] = -100                                   # the prompt head is not supervised (like the SFT collator)
    labels[~mask] = -100
    images = np.clip(rs.standard_normal((B, 3, 224, 224)), -2.5, 2.5).astype(np.float32)
    return dict(input_ids=ids, attention_mask=mask, labels=labels, images=images)


def build_reference(w):
    from transformers import CLIPImageProcessor, CLIPVisionConfig, CLIPVisionModel, Qwen2Config
    from dexbotic.model.dexbotic_arch import DexboticConfig, DexboticForCausalLM
    cfg = REAL4_LM
    d = os.path.join(tempfile.mkdtemp(), "clip_l")
    vcfg = CLIPVisionConfig(hidden_size=cfg.v_hidden, intermediate_size=cfg.v_inter, num_hidden_layers=cfg.v_layers,
                            num_attention_heads=cfg.v_heads, image_size=cfg.v_image, patch_size=cfg.v_patch,
                            layer_norm_eps=cfg.v_eps)
    CLIPVisionModel(vcfg).save_pretrained(d)
    CLIPImageProcessor(size={"shortest_edge": cfg.v_image}, crop_size={"height": cfg.v_image, "width": cfg.v_image}).save_pretrained(d)
    llm = Qwen2Config(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                      num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                      num_key_value_heads=cfg.num_key_value_heads, max_position_embeddings=4096, rope_theta=cfg.rope_theta,
                      rms_norm_eps=cfg.rms_norm_eps)
    m = DexboticForCausalLM(DexboticConfig(llm_config=llm, mm_vision_tower=d, mm_projector_type="mlp2x_gelu"))
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == {k: v.shape for k, v in w.items()}
    m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=True)
    for p_ in m.parameters():
        p_.requires_grad = True
    return m


def positions(x):
    """logit rows kept per sample: every POS_STRIDE-th position and the last valid one"""
    S = 256 + x["input_ids"].shape[1] - 1
    keep = sorted(set(range(0, S, POS_STRIDE)) | {S - 1})
    return np.array(keep, dtype=np.int64)


def step(m, x, autocast: bool):
    t = torch.from_numpy
    m.train()
    m.zero_grad(set_to_none=True)
    kw = dict(input_ids=t(x["input_ids"]), attention_mask=t(x["attention_mask"]), labels=t(x["labels"]), images=t(x["images"]))
    if autocast:
        with torch.autocast("cpu", dtype=torch.bfloat16):
            out = m(**kw)
    else:
        out = m(**kw)
    out.loss.backward()
    res = {"loss": np.float64(out.loss.item()), "logits": out.logits.detach().float().numpy()[:, positions(x)].astype(np.float32)}
    sd = dict(m.named_parameters())
    gsq = {g: 0.0 for g in GROUPS}
    for n, p_ in sd.items():
        if p_.grad is None:
            continue
        for g, pre in GROUPS.items():
            if n.startswith(pre):
                gsq[g] += float(p_.grad.double().pow(2).sum())
    for g in GROUPS:
        res["gnorm/" + g] = np.float64(gsq[g] ** 0.5)
    for n in GSAMP:
        res["gsamp/" + n] = sd[n].grad.detach().float().reshape(-1)[::STRIDE].numpy().astype(np.float32)
    return res


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


def main():
    sys.path.insert(0, G.REF)
    sys.path.insert(0, G.ROOT)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    G.install_timm_shim()
    t0 = time.time()
    w = lm_weights()
    x = inputs()
    m = build_reference(w)
    out = dict(weights_crc=np.uint32(weights_crc(w)), seed=np.int64(SEED), positions=positions(x), **x)
    runs = {}
    for tag, ac in (("fp32", False), ("bf16", True)):
        runs[tag] = step(m, x, ac)
        for k, v in runs[tag].items():
            out[f"{tag}/{k}"] = v
        print(f"[lm_real] {tag}: loss {runs[tag]['loss']:.6f} gnorm " +
              " ".join(f"{g} {runs[tag]['gnorm/' + g]:.4g}" for g in GROUPS) + f"  ({time.time() - t0:.0f} s)", flush=True)
    for k in runs["fp32"]:
        out["ref_bf16_vs_fp32/" + k] = np.float64(rel(runs["bf16"][k], runs["fp32"][k]))
    # greedy continuation of sample 0 by full-prefix recompute (fp32)
    m.eval()
    t = torch.from_numpy
    cur = t(x["input_ids"][:1]).clone()
    img1 = t(x["images"][:1])
    new, rows = [], []
    with torch.no_grad():
        for _ in range(N_NEW):
            lg = m(input_ids=cur, images=img1).logits[0, -1].float()
            nxt = int(torch.argmax(lg))
            new.append(nxt)
            rows.append(lg.numpy().astype(np.float32))
            cur = torch.cat([cur, torch.tensor([[nxt]], dtype=cur.dtype)], dim=1)
    out["decode_prompt"] = x["input_ids"][:1]
    out["decode_new_ids"] = np.array(new, dtype=np.int64)
    out["decode_logits"] = np.stack(rows)
    top2 = np.sort(out["decode_logits"], axis=1)[:, -2:]
    out["decode_margin"] = (top2[:, 1] - top2[:, 0]).astype(np.float32)
    np.savez_compressed(os.path.join(G.GOLD, "lm_real_ref.npz"), **out)
    print(f"[lm_real] greedy ids {new} margins {out['decode_margin'].round(4).tolist()}  total {time.time() - t0:.0f} s", flush=True)
    print("[lm_real] reference bf16 vs fp32:", {k: round(float(out['ref_bf16_vs_fp32/' + k]), 6) for k in ("loss", "logits", "gnorm/llm")})


if __name__ == "__main__":
    main()
