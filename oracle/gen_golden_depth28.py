"""Golden vectors at the depth bench.py times: the REFERENCE'S OWN CogACTForCausalLM at the BASELINE widths, **28 decoder
layers + the full CLIP-L tower (24 layers, hidden_states[-2])**, forward only, for the p50 workload of BASELINE.json configs[1]
(batch 1, two 224 x 224 views, 32 prompt tokens -> S = 543, CFG 1.5, 10 DDIM steps):

    python -m oracle.gen_golden_depth28          # ~15-25 min on 8 cores, peak ~45 GB -> tests/golden/cogact_depth28_ref.npz

Two runs of ``inference_action`` on the same weights and inputs:
  * "bf16": every parameter in bfloat16, exactly what ``CogACTForCausalLM.from_pretrained(..., torch_dtype=torch.bfloat16)``
    (dexbotic/exp/cogact_exp.py:134-138) hands to ``inference_action`` (cogact_arch.py:149-198): bf16 tower, decoder AND DiT;
    the DDIM state is fp32 (the float64 tables make it so, diffusion.py:984) and is cast to bf16 at every DiT call (dit.py:301);
  * "fp32": the SAME bf16-rounded values upcast to float32 and run in float32 — the exact arithmetic on those weights.  The
    distance between the two is the reference's OWN bf16 error at depth 28; the product's bf16 request is held to the fp32 run
    within a multiple of it and to the bf16 run at a stated bound (tests/test_depth28_gpu.py).
Weights: oracle/weights.fast_weight (PCG64 stream per tensor, depth-scaled residual branches), regenerated from the seed on the
test side; a strided CRC pins them.  Stored: cognition feature, the DDIM result (both CFG halves' first half = the returned
chunk before de-normalisation), the de-normalised chunk, and the distance of the ORACLE (fp32, on the same weights) to the fp32
run: the oracle is held to the reference at depth 28 HERE (29 GB of weights do not belong in the CPU test suite; the CPU test
checks the recorded distances and re-runs them when DXA_HEAVY_TESTS=1).
TEST INFRASTRUCTURE: runs only in the build container (needs /root/reference)."""
import os
import sys
import time
import zlib

import numpy as np
import torch

from . import cogact_oracle as O
from . import gen_golden as G
from .weights import cogact_shapes, fast_sample_crc, fast_weight_items

REAL28 = O.OracleConfig(vocab_size=2048, hidden_size=3584, intermediate_size=18944, num_hidden_layers=28,
                        num_attention_heads=28, num_key_value_heads=4, v_hidden=1024, v_inter=4096, v_layers=24, v_heads=16,
                        v_image=224, v_patch=14, dit_hidden=768, dit_depth=12, dit_heads=12)
SEED = 41
NORMS = {"min": [-1.0, -0.5, -2.0, -1.0, -1.0, -3.0, 0.0], "max": [1.0, 0.5, 2.0, 1.0, 1.0, 3.0, 1.0]}


def inputs():
    rs = np.random.RandomState(19)
    ids = rs.randint(10, REAL28.vocab_size, size=(1, 32)).astype(np.int64)
    ids[:, 1] = -200
    images = np.clip(rs.standard_normal((1, 2, 3, 224, 224)), -2.5, 2.5).astype(np.float32)   # B = 1, 2 views
    init = rs.standard_normal((1, 16, 7)).astype(np.float32)
    return dict(infer_ids=ids, infer_images=images, infer_init=init)


def bf16_round(a: np.ndarray) -> torch.Tensor:
    """fp32 values of the bf16-rounded tensor (torch's round-to-nearest-even)"""
    return torch.from_numpy(a).bfloat16().float()


def run_inference(m, x, dtype):
    t = torch.from_numpy
    m.eval()
    head = m.model.action_head
    if head.ddim_diffusion is None:
        head.create_ddim(ddim_step=10)
    dd = head.ddim_diffusion
    orig_loop = dd.ddim_sample_loop
    got = {}

    def loop(fn, shape, noise, **k):
        got["z"] = k["model_kwargs"]["z"][:1].detach().float().clone()
        s = orig_loop(fn, shape, noise, **k)
        got["samples"] = s.detach().float().clone()
        return s
    dd.ddim_sample_loop = loop
    try:
        with G.inject_rng(init_noise=t(x["infer_init"])), torch.no_grad():
            acts = m.inference_action(t(x["infer_ids"]), t(x["infer_images"]).to(dtype),
                                      {"cfg_scale": 1.5, "num_ddim_steps": 10, "action_norms": NORMS})
    finally:
        dd.ddim_sample_loop = orig_loop
    return {"infer_cognition": got["z"].numpy(), "infer_samples": got["samples"][:1].numpy(),
            "actions": np.asarray(acts, dtype=np.float64)}


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


def main():
    global REAL28
    dry = int(os.environ.get("DXA_D28_DRY_LAYERS", "0"))       # mechanics check of this script at a small depth (writes to /tmp)
    if dry:
        import dataclasses
        REAL28 = dataclasses.replace(REAL28, num_hidden_layers=dry, v_layers=3)
    torch.set_num_threads(os.cpu_count() or 8)
    sys.path.insert(0, G.REF)
    G.install_timm_shim()
    t0 = time.time()
    shapes = cogact_shapes(REAL28)
    torch.set_default_dtype(torch.bfloat16)          # what from_pretrained(torch_dtype=bfloat16) does while it builds the model
    try:
        m = G.build_reference(REAL28, None)
    finally:
        torch.set_default_dtype(torch.float32)
    m.model.mm_vision_tower.to(torch.bfloat16)       # the nested tower's checkpoint values arrive in the outer dtype too
    for p in m.parameters():
        p.requires_grad_(False)
    print(f"reference built ({sum(p.numel() for p in m.parameters()) / 1e9:.2f} B parameters) {time.time()-t0:.0f}s", flush=True)
    sd = m.state_dict()
    crc = 0
    with torch.no_grad():
        for name, arr in fast_weight_items(shapes, SEED, depth_scale=REAL28.num_hidden_layers):
            crc = fast_sample_crc(arr, crc)
            assert sd[name].dtype == torch.bfloat16, (name, sd[name].dtype)
            sd[name].copy_(torch.from_numpy(arr))    # fp32 -> bf16, round to nearest even
    print(f"weights loaded, crc {crc} {time.time()-t0:.0f}s", flush=True)
    x = inputs()
    res = {"seed": np.int64(SEED), "weights_crc": np.int64(crc), "infer_ids": x["infer_ids"], "infer_init": x["infer_init"],
           "infer_images_crc": np.int64(zlib.crc32(x["infer_images"].tobytes())),
           "norm_min": np.asarray(NORMS["min"]), "norm_max": np.asarray(NORMS["max"])}
    r16 = run_inference(m, x, torch.bfloat16)
    print("reference bf16 done", r16["actions"][0], f"{time.time()-t0:.0f}s", flush=True)
    m.float()                                        # the same (bf16-rounded) values, float32 arithmetic
    r32 = run_inference(m, x, torch.float32)
    print("reference fp32 done", r32["actions"][0], f"{time.time()-t0:.0f}s", flush=True)
    for tag, r in (("bf16", r16), ("fp32", r32)):
        for k, v in r.items():
            res[f"{tag}/{k}"] = np.asarray(v)
    for k in r32:
        res[f"ref_bf16_vs_fp32/{k}"] = np.float64(rel(r16[k], r32[k]))
    print("reference bf16 vs fp32", {k: f"{rel(r16[k], r32[k]):.2e}" for k in r32}, flush=True)
    # the oracle on the same float32 weights (views of the reference's storage: no second copy)
    osd = {k: v for k, v in m.state_dict().items()}
    t = torch.from_numpy
    with torch.no_grad():
        io = O.cogact_forward(osd, REAL28, t(x["infer_ids"]), None, t(x["infer_images"]))
        cog1 = io["logits"][:, -1, :][:, None, :].float()
        samples = O.ddim_sample(osd, REAL28, cog1, t(x["infer_init"]), 1.5, 10)
    od = {"infer_cognition": rel(cog1.numpy(), r32["infer_cognition"]), "infer_samples": rel(samples.numpy(), r32["infer_samples"])}
    print("oracle vs reference fp32", {k: f"{v:.2e}" for k, v in od.items()}, f"{time.time()-t0:.0f}s", flush=True)
    for k, v in od.items():
        res[f"oracle_vs_ref/fp32/{k}"] = np.float64(v)
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "cogact_depth28_ref.npz")
    if dry:
        dst = "/tmp/cogact_depth28_dry.npz"
    np.savez_compressed(dst, **res)
    print("wrote", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
