"""Two deeper pins of the TRAINING step from the REFERENCE'S OWN CLASSES (round 5, verdict item 4):

    python -m oracle.gen_golden_traj traj      # ~25 min on 8 cores  -> tests/golden/cogact_traj_ref.npz
    python -m oracle.gen_golden_traj depth12   # ~40 min, peak ~45 GB -> tests/golden/cogact_depth12_ref.npz

"traj": dexbotic.model.cogact.cogact_arch.CogACTForCausalLM at the BASELINE widths with 4 decoder layers (the model, weights
and batch of oracle/gen_golden_realwidth_ref.py) driven through FIVE optimizer steps of what HF ``Trainer`` does for the
reference's recipe (dexbotic/exp/trainer.py:25-36,88-124; base_exp.py:95-203): forward under ``torch.autocast(bfloat16)``
(and, second run, plain fp32) -> backward -> ``clip_grad_norm_(1.0)`` -> ``torch.optim.AdamW`` with the reference's decay /
no-decay grouping (lr 2e-5 = the recipe's base_lr, betas (0.9, 0.999), eps 1e-8, weight decay 0.01, constant lr).  Every step sees the same episodes with
FRESH injected draws (noise, timesteps, condition drop).  Stored per run: the five losses, the five pre-clip gradient norms and
strided samples of (parameters after step 5 - initial parameters) for ten tensors; plus the distance between the reference's own
bf16 and fp32 trajectories, which is the yardstick for the product's bf16 bound.

"depth12": the same model with TWELVE decoder layers (2.85 B parameters; a depth-28 forward + backward of the reference does not
fit this container's 62 GB, depth 12 does), one step, fp32 and bf16 autocast: loss, cognition feature, per-group gradient norms,
strided gradient samples.  Weights from oracle/weights.fast_weight_items (one PCG64 stream per tensor, regenerated on the test
side; a strided CRC pins them).
TEST INFRASTRUCTURE: runs only in the build container (needs /root/reference)."""
import dataclasses
import os
import sys
import time
import zlib

import numpy as np
import torch

from . import gen_golden as G
from . import gen_golden_realwidth_ref as R
from .weights import cogact_shapes, fast_sample_crc, fast_weight_items, make_weights, weights_crc

STEPS = 5
LR, WD = 2e-5, 0.01           # the recipe's base_lr (dexbotic/exp/base_exp.py: OptimizerConfig.base_lr = 2e-5)
DSTRIDE = 1009
REAL12 = dataclasses.replace(R.REAL4, num_hidden_layers=12)
SEED12 = 53
GSAMP12 = ("model.llm.layers.0.self_attn.q_proj.weight", "model.llm.layers.11.mlp.down_proj.weight",
           "model.llm.layers.6.mlp.gate_proj.weight", "model.llm.layers.3.self_attn.o_proj.weight",
           "model.llm.layers.0.self_attn.k_proj.bias", "model.llm.layers.11.input_layernorm.weight",
           "model.mm_projector.2.weight", "model.mm_vision_tower.vision_tower.encoder.layers.0.mlp.fc1.weight",
           "model.action_head.net.blocks.11.mlp.fc2.weight", "model.action_head.net.z_embedder.linear.weight")


def step_draws(step: int, B: int = 2):
    """the injected draws of optimizer step ``step`` (0-based): [4 B, 16, 7] noise, [4 B] timesteps, [4 B] drop uniforms"""
    rs = np.random.RandomState(1000 + step)
    noise = rs.standard_normal((4 * B, 16, 7)).astype(np.float32)
    ts = rs.randint(0, 100, size=(4 * B,)).astype(np.int64)
    drop_u = rs.uniform(size=(4 * B,)).astype(np.float32)
    drop_u[step % (4 * B)] = 0.01                         # at least one dropped condition per step
    return noise, ts, drop_u


def fwd_bwd(m, x, noise, ts, drop_u, autocast: bool):
    t = torch.from_numpy
    m.train()
    with G.inject_rng(noise=t(noise), timesteps=t(ts), drop_u=t(drop_u)), R.fp32_head(m):
        with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
            out = m(input_ids=t(x["input_ids"]), attention_mask=t(x["attention_mask"]), images=t(x["images"]),
                    actions=t(x["actions"]), labels=t(x["input_ids"]).clone())
    out.loss.backward()
    return out


def run_traj(m, x, autocast: bool, names):
    named = [(n, p) for n, p in m.named_parameters() if p.requires_grad]
    p0 = {n: dict(named)[n].detach().reshape(-1)[::DSTRIDE].clone() for n in names}
    opt = None
    losses, norms = [], []
    for s in range(STEPS):
        noise, ts, du = step_draws(s)
        m.zero_grad(set_to_none=True)
        out = fwd_bwd(m, x, noise, ts, du, autocast)
        with_grad = [(n, p) for n, p in named if p.grad is not None]
        if opt is None:
            groups = [dict(params=[p for n, p in with_grad if not G.no_decay_name(n)], weight_decay=WD),
                      dict(params=[p for n, p in with_grad if G.no_decay_name(n)], weight_decay=0.0)]
            opt = torch.optim.AdamW(groups, lr=LR, betas=(0.9, 0.999), eps=1e-8)
        total = torch.nn.utils.clip_grad_norm_([p for _, p in with_grad], 1.0)
        opt.step()
        losses.append(float(out.loss.item()))
        norms.append(float(total))
    pd = {n: (dict(named)[n].detach().reshape(-1)[::DSTRIDE] - p0[n]).float().numpy().copy() for n in names}
    return {"losses": np.asarray(losses), "norms": np.asarray(norms), **{"delta/" + n: v for n, v in pd.items()}}


def oracle_step(sd, cfg, x, noise, ts, drop_u, autocast: bool):
    """the CPU restatement's forward + backward on the same inputs and draws (oracle/cogact_oracle.py)"""
    from . import cogact_oracle as O
    t = torch.from_numpy
    with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
        feats = O.extract_vision_features(sd, cfg, t(x["images"]))
        src, new_mask, _ = O.splice_plan(x["input_ids"], x["attention_mask"], feats.shape[1], None, "right")
        hidden = O.qwen2_forward(sd, cfg, O.splice_embeds(sd, src, feats.float()), t(new_mask))
        cog = O.cognition_features(hidden, t(new_mask))
    with torch.autocast("cpu", enabled=False):
        loss, _, _ = O.action_loss(sd, cfg, t(x["actions"]), cog.float(), t(noise), t(ts), t(drop_u) < 0.1, 4)
    loss.backward()
    return loss, cog


def run_traj_oracle(w, x, autocast: bool, names, steps=STEPS):
    """the same five optimizer steps driven through the ORACLE (its tensors as torch.optim.AdamW parameters)"""
    sd = {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in w.items()}
    p0 = {n: sd[n].detach().reshape(-1)[::DSTRIDE].clone() for n in names}
    opt, losses, norms = None, [], []
    for s in range(steps):
        noise, ts, du = step_draws(s)
        for p in sd.values():
            p.grad = None
        loss, _ = oracle_step(sd, R.REAL4, x, noise, ts, du, autocast)
        with_grad = [(n, p) for n, p in sd.items() if p.grad is not None]
        if opt is None:
            opt = torch.optim.AdamW([dict(params=[p for n, p in with_grad if not G.no_decay_name(n)], weight_decay=WD),
                                     dict(params=[p for n, p in with_grad if G.no_decay_name(n)], weight_decay=0.0)],
                                    lr=LR, betas=(0.9, 0.999), eps=1e-8)
        norms.append(float(torch.nn.utils.clip_grad_norm_([p for _, p in with_grad], 1.0)))
        opt.step()
        losses.append(float(loss.item()))
    pd = {n: (sd[n].detach().reshape(-1)[::DSTRIDE] - p0[n]).float().numpy().copy() for n in names}
    return {"losses": np.asarray(losses), "norms": np.asarray(norms), **{"delta/" + n: v for n, v in pd.items()}}


def traj_dist(got, ref):
    out = {}
    for k in ref:
        a, b = np.asarray(got[k], dtype=np.float64), np.asarray(ref[k], dtype=np.float64)
        out[k] = float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30)) if k.startswith("delta/") else \
            float(np.abs(a - b).max() / np.abs(b).max())
    return out


def main_traj():
    t0 = time.time()
    w = make_weights(cogact_shapes(R.REAL4), R.SEED)
    x = R.inputs()
    res = {"seed": np.int64(R.SEED), "weights_crc": np.int64(weights_crc(w)), "steps": np.int64(STEPS),
           "lr": np.float64(LR), "weight_decay": np.float64(WD), "dstride": np.int64(DSTRIDE),
           "images_crc": np.int64(zlib.crc32(x["images"].tobytes()))}
    runs = {}
    for tag, ac in (("fp32", False), ("bf16", True)):
        m = G.build_reference(R.REAL4, w)                 # fresh masters for each run
        r = runs[tag] = run_traj(m, x, ac, R.GSAMP)
        del m
        print("reference", tag, "losses", np.round(r["losses"], 5), "norms", np.round(r["norms"], 4), f"{time.time()-t0:.0f}s",
              flush=True)
        for k, v in r.items():
            res[f"{tag}/{k}"] = v
    for k in runs["fp32"]:
        a, b = runs["bf16"][k].astype(np.float64), runs["fp32"][k].astype(np.float64)
        if k.startswith("delta/"):
            res["ref_bf16_vs_fp32/" + k] = np.float64(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))
        else:
            res["ref_bf16_vs_fp32/" + k] = np.float64(np.abs(a - b).max() / np.abs(b).max())
    print({k: f"{float(v):.2e}" for k, v in res.items() if k.startswith("ref_bf16_vs_fp32/")}, flush=True)
    # the ORACLE through the same five steps (fp32): its distance to the reference's trajectory is recorded here and re-checked by
    # tests/test_oracle_realwidth.py (re-run there only with DXA_HEAVY_TESTS=1: ten minutes of CPU)
    d = traj_dist(run_traj_oracle(w, x, False, R.GSAMP), runs["fp32"])
    print("oracle vs reference fp32 trajectory", {k[-40:]: f"{v:.2e}" for k, v in d.items()}, f"{time.time()-t0:.0f}s", flush=True)
    for k, v in d.items():
        res["oracle_vs_ref/fp32/" + k] = np.float64(v)
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "cogact_traj_ref.npz")
    np.savez_compressed(dst, **res)
    print("wrote", dst, os.path.getsize(dst), "bytes")


def load_fast(sd, shapes, seed, depth):
    crc = 0
    with torch.no_grad():
        for name, arr in fast_weight_items(shapes, seed, depth_scale=depth):
            crc = fast_sample_crc(arr, crc)
            sd[name].copy_(torch.from_numpy(arr))
    return crc


def run_depth12(m, x, autocast: bool, cfg):
    m.zero_grad(set_to_none=True)
    out = fwd_bwd(m, x, x["noise"], x["timesteps"], x["drop_u"], autocast)
    hid = out.logits.detach().float()
    lens = x["attention_mask"].sum(1) - 1 + cfg.num_patches
    cog = torch.stack([hid[b, int(lens[b]) - 1] for b in range(hid.shape[0])])[:, None, :]
    res = {"loss": np.float64(out.loss.item()), "cognition": cog.numpy()}
    grads = {n: p.grad for n, p in m.named_parameters()}
    for g, pre in R.GROUPS.items():
        sq = sum(float(v.double().pow(2).sum()) for n, v in grads.items() if n.startswith(pre) and v is not None)
        res[f"gnorm/{g}"] = np.float64(sq ** 0.5)
    for n in GSAMP12:
        if n in grads:                                    # (a dry run at a smaller depth lacks the deep layers)
            res["gsamp/" + n] = grads[n].reshape(-1)[::R.STRIDE].float().numpy().copy()
    m.zero_grad(set_to_none=True)
    return res


def main_depth12():
    t0 = time.time()
    cfg = REAL12
    dry = int(os.environ.get("DXA_D12_DRY_LAYERS", "0"))
    if dry:
        cfg = dataclasses.replace(cfg, num_hidden_layers=dry)
    shapes = cogact_shapes(cfg)
    m = G.build_reference(cfg, None)
    crc = load_fast(m.state_dict(), shapes, SEED12, cfg.num_hidden_layers)
    print(f"reference built + weights loaded ({sum(p.numel() for p in m.parameters()) / 1e9:.2f} B parameters), crc {crc} "
          f"{time.time()-t0:.0f}s", flush=True)
    x = R.inputs()
    res = {"seed": np.int64(SEED12), "weights_crc": np.int64(crc), "layers": np.int64(cfg.num_hidden_layers),
           "images_crc": np.int64(zlib.crc32(x["images"].tobytes()))}
    res.update({k: v for k, v in x.items() if k not in ("images", "infer_images", "infer_ids", "infer_init")})
    runs = {}
    for tag, ac in (("fp32", False), ("bf16", True)):
        r = runs[tag] = run_depth12(m, x, ac, cfg)
        print("reference", tag, "loss", r["loss"], {k: round(float(v), 5) for k, v in r.items() if k.startswith("gnorm/")},
              f"{time.time()-t0:.0f}s", flush=True)
        for k, v in r.items():
            res[f"{tag}/{k}"] = np.asarray(v)
    for k in runs["fp32"]:
        res["ref_bf16_vs_fp32/" + k] = np.float64(R.rel(runs["bf16"][k], runs["fp32"][k]))
    print({k: f"{float(v):.2e}" for k, v in res.items() if k.startswith("ref_bf16_vs_fp32/")}, flush=True)
    # the ORACLE at depth 12 on the reference's own fp32 storage (views: no second copy of 11 GB)
    sd = {k: v.detach().requires_grad_(True) for k, v in m.state_dict().items()}
    del m
    loss, cog = oracle_step(sd, cfg, x, x["noise"], x["timesteps"], x["drop_u"], False)
    od = {"loss": R.rel(loss.item(), runs["fp32"]["loss"]), "cognition": R.rel(cog.detach().float().numpy(), runs["fp32"]["cognition"])}
    for g_, pre in R.GROUPS.items():
        sq = sum(float(p.grad.double().pow(2).sum()) for n, p in sd.items() if n.startswith(pre) and p.grad is not None)
        od[f"gnorm/{g_}"] = R.rel(sq ** 0.5, runs["fp32"][f"gnorm/{g_}"])
    for n in GSAMP12:
        if n in sd and sd[n].grad is not None:
            od["gsamp/" + n] = R.rel(sd[n].grad.reshape(-1)[::R.STRIDE].float().numpy(), runs["fp32"]["gsamp/" + n])
    print("oracle vs reference fp32 at depth 12", {k[-40:]: f"{v:.2e}" for k, v in od.items()}, f"{time.time()-t0:.0f}s", flush=True)
    for k, v in od.items():
        res["oracle_vs_ref/fp32/" + k] = np.float64(v)
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "cogact_depth12_ref.npz")
    if dry:
        dst = "/tmp/cogact_depth12_dry.npz"
    np.savez_compressed(dst, **res)
    print("wrote", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    torch.set_num_threads(os.cpu_count() or 8)
    sys.path.insert(0, G.REF)
    G.install_timm_shim()
    what = sys.argv[1] if len(sys.argv) > 1 else "traj"
    {"traj": main_traj, "depth12": main_depth12}[what]()
