#!/usr/bin/env python
"""DB-CogACT fine-tune throughput on MI355X (BASELINE.json metric: episodes/sec @1/2/4/8 GPU + p50
action-inference ms).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A "step" = one optimizer step of the native policy (dexbotic_amd) on a synthetic batch of `--batch`
episodes per GPU (default 16 => global 128 at DP=8, BASELINE config "batch 128 synthetic episodes"):
ViT (CLIP-L/14 @224) -> mlp2x_gelu projector -> splice -> Qwen2.5-7B-class decoder (28 L) -> cognition token
-> DiT-B diffusion loss (4 repeats) -> backward -> [RCCL gradient all-reduce] -> global-norm clip -> fused AdamW.
Random-init weights at the REAL shapes (no checkpoints offline), bf16 compute with fp32 master weights,
fp32 action head; nothing is skipped inside the timed region.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0   # dense MFMA bf16, MI355X_MICROARCH.md chip table


def flops_per_sample_fwd(llm, vis, V, s_text, dit_hidden, dit_depth, T, R):
    """GEMM = 2*params*tokens, attention = 4*S^2*d per layer (full, not causal-halved)."""
    np_ = (vis["image_size"] // vis["patch_size"]) ** 2
    C, I, Lv = vis["hidden_size"], vis["intermediate_size"], vis["num_hidden_layers"]
    d, f, Ll = llm["hidden_size"], llm["intermediate_size"], llm["num_hidden_layers"]
    Hq, Hkv = llm["num_attention_heads"], llm["num_key_value_heads"]
    hd = d // Hq
    S = s_text - 1 + V * np_
    vit_layer = 4 * C * C + 2 * C * I
    f_vit = V * (Lv * (2 * vit_layer * (np_ + 1) + 4 * (np_ + 1) ** 2 * C) + 2 * (3 * 14 * 14 * C) * np_)
    f_proj = 2 * (C * d + d * d) * V * np_
    llm_layer = d * (Hq + 2 * Hkv) * hd + Hq * hd * d + 3 * d * f
    f_llm = Ll * (2 * llm_layer * S + 4 * S * S * d)
    h = dit_hidden
    f_dit = R * (dit_depth * (2 * 12 * h * h * (T + 1) + 4 * (T + 1) ** 2 * h) + 2 * (d * h + 256 * h + h * h))
    return f_vit + f_proj + f_llm + f_dit, S


def build_model(args, device):
    from dexbotic_amd.model.cogact.cogact_arch import CogActConfig, CogACTForCausalLM
    from dexbotic_amd.model.llm.qwen2 import Qwen2Config
    from dexbotic_amd.model.modules.mm_vision.clip.clip_encoder import CLIPVisionConfig
    llm = Qwen2Config(num_hidden_layers=args.llm_layers)          # Qwen2.5-7B shapes
    vis = CLIPVisionConfig(num_hidden_layers=args.vit_layers)    # CLIP ViT-L/14 @224
    cfg = CogActConfig(llm_config=llm, mm_vision_tower=vis, mm_projector_type="mlp2x_gelu", action_model_type="DiT-B",
                       action_dim=7, chunk_size=16, compute_dtype=args.dtype)
    model = CogACTForCausalLM(cfg, device=device, train=True)
    model.init_random_(seed=0)
    return model, cfg, llm, vis


def synthetic_batch(B, V, s_text, device, seed):
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    shape = (B, V, 3, 224, 224) if V > 1 else (B, 3, 224, 224)
    images = torch.randn(shape, generator=g).clamp_(-2.5, 2.5)
    ids = torch.randint(1000, 30000, (B, s_text), generator=g)
    ids[:, 1] = -200
    actions = torch.rand(B, 112, generator=g) * 2 - 1
    return dict(input_ids=ids.to(device), attention_mask=torch.ones(B, s_text, dtype=torch.bool, device=device),
                images=images.to(device), actions=actions.to(device), labels=ids.to(device))


def host_batches(n: int, B: int, V: int, s_text: int, seed: int, ragged: bool = False):
    """n distinct COLLATED batches on the host, as a DataLoader(pin_memory=True) hands them over: float inputs pinned, integer
    inputs plain host tensors.  ``ragged``: every fourth batch carries right-padded instructions (lengths 20..s_text, at least one
    full-length sample so S stays the BASELINE 287)."""
    out = []
    for k in range(n):
        b = synthetic_batch(B, V, s_text, "cpu", seed + 7919 * k)
        if ragged and k % 4 == 3:
            g = torch.Generator().manual_seed(seed + k)
            lens = torch.randint(20, s_text + 1, (B,), generator=g)
            lens[0] = s_text
            b["attention_mask"] = torch.arange(s_text)[None, :] < lens[:, None]
        for key in ("images", "actions"):
            b[key] = b[key].pin_memory()
        out.append(b)
    return out


def rotating_batches(batches, seed: int):
    """endless stream over ``batches`` with FRESH token ids every step (a real fine-tune never sees the same instruction
    batch twice: the splice plan is rebuilt and uploaded every step, nothing is served from the plan cache)"""
    rng = np.random.default_rng(seed)
    k = 0
    while True:
        b = dict(batches[k % len(batches)])
        ids = torch.from_numpy(rng.integers(1000, 30000, size=tuple(b["input_ids"].shape), dtype=np.int64))
        ids[:, 1] = -200
        b["input_ids"], b["labels"] = ids, ids
        k += 1
        yield b


def _time_steps(step, warm: int, n_max: int, budget_s: float):
    """median seconds per call of ``step`` over up to n_max timed calls (at least 2) within ~budget_s, after `warm` calls"""
    for _ in range(warm):
        step()
    ts = []
    t_start = time.perf_counter()
    while len(ts) < 2 or (len(ts) < n_max and time.perf_counter() - t_start < budget_s):
        t0 = time.perf_counter()
        step()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), len(ts)


def _cpu_port_step(cfg_llm, cfg_vis, L_llm, L_vit, L_dit, B, args):
    """one fine-tune step (fwd + bwd + AdamW) of the CPU oracle (oracle/cogact_oracle.py, a port of the reference
    algorithm) at the REAL widths with L_llm decoder layers"""
    from oracle import cogact_oracle as O
    from oracle.weights import cogact_shapes
    oc = O.OracleConfig(vocab_size=cfg_llm.vocab_size, hidden_size=cfg_llm.hidden_size,
                        intermediate_size=cfg_llm.intermediate_size, num_hidden_layers=L_llm,
                        num_attention_heads=cfg_llm.num_attention_heads,
                        num_key_value_heads=cfg_llm.num_key_value_heads, v_hidden=cfg_vis.hidden_size,
                        v_inter=cfg_vis.intermediate_size, v_layers=L_vit + 1, v_heads=cfg_vis.num_attention_heads,
                        v_image=cfg_vis.image_size, v_patch=cfg_vis.patch_size, dit_hidden=768, dit_depth=L_dit,
                        dit_heads=12)
    shapes = cogact_shapes(oc)
    shapes.pop("lm_head.weight")
    g = torch.Generator().manual_seed(0)
    sd = {k: (torch.randn(s, generator=g) * 0.02).requires_grad_(True) for k, s in shapes.items()}
    batch = synthetic_batch(B, args.views, args.s_text, "cpu", 1)
    noise = torch.randn(4 * B, 16, 7, generator=g)
    ts = torch.randint(0, 100, (4 * B,), generator=g)
    params = list(sd.values())
    state = {}

    def step():
        out = O.cogact_forward(sd, oc, batch["input_ids"], batch["attention_mask"], batch["images"], batch["actions"],
                               noise, ts, None)
        out["loss"].backward()
        with torch.no_grad():
            ps = [p for p in params if p.grad is not None]
            for p in ps:
                if id(p) not in state:
                    state[id(p)] = (torch.zeros_like(p), torch.zeros_like(p))
            O.adamw_step(ps, [p.grad for p in ps], [state[id(p)][0] for p in ps], [state[id(p)][1] for p in ps],
                         1, 2e-5)
        for p in params:
            p.grad = None
    return step


def _cpu_reference_step(cfg_llm, cfg_vis, L_llm, L_vit, L_dit, B, args):
    """the same step through the REFERENCE's own classes (dexbotic.model.cogact.cogact_arch.CogACTForCausalLM +
    torch.optim.AdamW), imported from /root/reference with the shims of oracle/gen_golden.py — only where that tree
    exists (the build container); the GPU box times the port"""
    from oracle import cogact_oracle as O
    from oracle import gen_golden as G
    sys.path.insert(0, G.REF)
    G.install_timm_shim()
    oc = O.OracleConfig(vocab_size=cfg_llm.vocab_size, hidden_size=cfg_llm.hidden_size,
                        intermediate_size=cfg_llm.intermediate_size, num_hidden_layers=L_llm,
                        num_attention_heads=cfg_llm.num_attention_heads,
                        num_key_value_heads=cfg_llm.num_key_value_heads, v_hidden=cfg_vis.hidden_size,
                        v_inter=cfg_vis.intermediate_size, v_layers=L_vit + 1, v_heads=cfg_vis.num_attention_heads,
                        v_image=cfg_vis.image_size, v_patch=cfg_vis.patch_size, dit_hidden=768, dit_depth=L_dit,
                        dit_heads=12)
    m = G.build_reference(oc, None)
    m.train()
    opt = torch.optim.AdamW([p for n, p in m.named_parameters() if p.requires_grad and not n.startswith("lm_head")],
                            lr=2e-5, weight_decay=0.0)
    batch = synthetic_batch(B, args.views, args.s_text, "cpu", 1)

    def step():
        out = m(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], labels=batch["labels"],
                images=batch["images"], actions=batch["actions"])
        out.loss.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
        opt.step()
        opt.zero_grad(set_to_none=True)
    return step


def cpu_baseline(args, cfg_llm, cfg_vis):
    """Reported baseline (never the target): the reference algorithm on the host cores, fp32, fwd + bwd + AdamW at the REAL
    widths.  A full 28-layer 7 B step on CPU takes minutes, so a bounded sample is timed (BASELINE.md §2 protocol:
    torch.set_num_threads(all physical cores), 2 warm-ups, median of up to 5 timed steps): decoder depths 1, 2 and 4 are
    timed, the least-squares line through them gives the marginal cost of a layer, and the full depth is extrapolated linearly
    in layer count (ViT layers by FLOP ratio to a decoder layer).  kind =
    "reference" when /root/reference is importable (its own CogACTForCausalLM + torch.optim.AdamW), else "port" (the CPU
    oracle, oracle/cogact_oracle.py)."""
    try:
        import psutil
        cores = psutil.cpu_count(logical=False) or os.cpu_count()
    except Exception:  # noqa: BLE001
        cores = os.cpu_count()
    B, L_vit, L_dit = 4, 2, 12
    use_ref = os.path.isdir("/root/reference/dexbotic") and not args.cpu_port
    make = _cpu_reference_step if use_ref else _cpu_port_step
    # thread count: more threads are not faster for these shapes on a 128-thread host (round 2: 128 threads delivered fewer
    # samples/s than 8 cores).  One depth-1 step is timed per candidate and the best count is used for everything after.
    cands = sorted({int(c) for c in (cores, cores // 2, cores // 4, 16, 8) if 1 <= int(c) <= cores})
    if args.cpu_threads:
        cands = [int(args.cpu_threads)]
    step1 = make(cfg_llm, cfg_vis, 1, L_vit, L_dit, B, args)
    torch.set_num_threads(max(cands))
    step1()                                              # warm-up (allocations, first-touch)
    sweep = {}
    for nt in cands:
        torch.set_num_threads(nt)
        t0 = time.perf_counter()
        step1()
        sweep[nt] = time.perf_counter() - t0
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    t1, n1 = _time_steps(step1, 0, 5, 1e9)               # >= 5 timed steps, median
    del step1
    t2, n2 = _time_steps(make(cfg_llm, cfg_vis, 2, L_vit, L_dit, B, args), 1, 3, 1e9)
    t4, n4 = _time_steps(make(cfg_llm, cfg_vis, 4, L_vit, L_dit, B, args), 1, 2, 1e9)
    # marginal cost of a decoder layer = slope of the least-squares line through the three depths (the 1 -> 2 difference
    # alone is two noisy ~15 s timings apart; the depth-4 point triples the lever arm); lin4 = what the 1 -> 2 difference alone
    # would have predicted for depth 4, reported beside the measurement
    xs, ts = (1.0, 2.0, 4.0), (t1, t2, t4)
    xm, tm = sum(xs) / 3, sum(ts) / 3
    per_layer = max(sum((x - xm) * (t - tm) for x, t in zip(xs, ts)) / sum((x - xm) ** 2 for x in xs), 1e-6)
    base = tm - per_layer * xm                       # everything that is not a decoder layer (ViT sample, DiT, embeddings, AdamW)
    lin4 = t1 + 3 * max(t2 - t1, 1e-6)
    # ViT layer / decoder layer forward-FLOP ratio (SURVEY.md section 8d formulas), per view
    npv = (cfg_vis.image_size // cfg_vis.patch_size) ** 2 + 1
    C, I = cfg_vis.hidden_size, cfg_vis.intermediate_size
    d, f = cfg_llm.hidden_size, cfg_llm.intermediate_size
    hd = d // cfg_llm.num_attention_heads
    S = args.s_text - 1 + args.views * (npv - 1)
    vit_layer = 2 * (4 * C * C + 2 * C * I) * npv + 4 * npv * npv * C
    llm_layer = 2 * (d * (cfg_llm.num_attention_heads + 2 * cfg_llm.num_key_value_heads) * hd + d * d + 3 * d * f) * S \
        + 4 * S * S * d
    est_full = base + per_layer * cfg_llm.num_hidden_layers \
        + per_layer * (vit_layer / llm_layer) * (cfg_vis.num_hidden_layers - 1 - L_vit) * args.views
    return {"value": round(B / est_full, 5), "unit": "episodes/s", "cores": int(best), "host_cores": int(cores),
            "thread_sweep_s_per_step_depth1": {str(k): round(v, 2) for k, v in sweep.items()},
            "kind": "reference" if use_ref else "port",
            "measured_4_layer_episodes_per_s": round(B / t4, 4), "linear_model_4_layer_episodes_per_s": round(B / lin4, 4),
            "sample": (f"{'reference CogACTForCausalLM + torch.optim.AdamW' if use_ref else 'CPU oracle (port)'}: fwd+bwd+AdamW, "
                       f"fp32, B={B} (GPU leg: {args.batch}; per-sample cost on the CPU is flat in B at these sizes), real "
                       f"widths, {L_vit} of 23 used ViT layers, DiT-B 12 layers, decoder depth 1 / 2 / 4 measured: "
                       f"{t1:.2f} s ({n1} steps) / {t2:.2f} s ({n2}) / {t4:.2f} s ({n4}) per step, median after warm-up, on the "
                       f"best of {sorted(sweep)} threads = {best}; "
                       f"full depth (28 decoder + 23 ViT layers) from the least-squares line through the three depths "
                       f"({per_layer:.2f} s per decoder layer)")}


def process_frame_latency(model, n_req: int = 50, views: int = 2) -> dict:
    """the wire format of BASELINE.json configs[0] / [1] (dexbotic/client.py:33-61 -> POST /process_frame,
    dexbotic/exp/base_exp.py:638-653, cogact_exp.py:146-177) through the Flask test client: multipart PNG frames + text ->
    PNG decode on the host -> pad / resize / crop / normalise on the device -> prompt -> inference_action -> JSON."""
    import io
    import types
    from dexbotic_amd.serve import InferenceServer, encode_png
    vocab = int(model.config.llm_config.vocab_size)

    class WordHashTokenizer:                       # no tokenizer files offline: one id per word (what travels is the id COUNT)
        bos_token_id = None

        def __call__(self, text):
            return types.SimpleNamespace(input_ids=[3 + (sum(map(ord, wd)) % (vocab - 3)) for wd in text.split()])

    srv = InferenceServer(model, WordHashTokenizer(), norm_stats={"min": [-1.0] * 7, "max": [1.0] * 7})
    client = srv.create_app().test_client()
    rs = np.random.RandomState(5)
    pngs = [encode_png(rs.randint(0, 256, (256, 256, 3)).astype(np.uint8)) for _ in range(8)]      # libero frames are 256 x 256
    texts = ["pick up the black bowl and place it on the plate", "open the top drawer and put the bowl inside",
             "turn on the stove and put the moka pot on it", "put the wine bottle on top of the cabinet"]
    inner = {}
    if os.environ.get("DXA_BENCH_PF_INNER"):           # tuning: time the pieces of inference_action (a device sync after each)
        from dexbotic_amd import kernels as K_

        def timed(obj, name, tag):
            fn = getattr(obj, name)

            def w(*a, **k):
                torch.cuda.synchronize()
                t_ = time.perf_counter()
                r_ = fn(*a, **k)
                torch.cuda.synchronize()
                inner.setdefault(tag, []).append(round(1e3 * (time.perf_counter() - t_), 2))
                return r_
            setattr(obj, name, w)
        timed(model, "_graph_sample", "graph_sample")
        timed(model.model._plans, "get", "plan")
        timed(K_, "dit_blocks_timed_out", "abort_word")
        timed(model, "_denorm", "denorm")
        timed(model.model.action_head.net, "refresh_packed", "refresh_packed")
    if os.environ.get("DXA_BENCH_PF_ONE_TEXT"):        # tuning: one prompt only
        texts = texts[:1]
    if os.environ.get("DXA_BENCH_PF_NOGC"):            # tuning: no cyclic garbage collection during the loop
        import gc
        gc.collect()
        gc.disable()
    from dexbotic_amd import hostcpu
    cg0 = hostcpu.throttle_stats()
    lat = []
    # warm-up: every prompt length three times — the request path keeps one captured HIP graph per sequence length (eager the first
    # time, captured the second, replayed from the third), and with 5 warm-up requests over 4 lengths the captures of three of them
    # fell INSIDE the timed requests (round 5: p90 56 ms against p50 22 ms)
    n_warm = 3 * len(texts)
    for i in range(n_warm + n_req):
        data = {"text": texts[i % len(texts)],
                "image": [(io.BytesIO(pngs[(i + v) % len(pngs)]), f"{v}.png") for v in range(views)]}
        t0 = time.perf_counter()
        r = client.post("/process_frame", content_type="multipart/form-data", data=data)
        lat.append(1e3 * (time.perf_counter() - t0))
        assert r.status_code == 200 and len(r.get_json()["response"]) == 16
    lat = np.asarray(lat[n_warm:])
    srv.model = None                                   # (the Flask app's closures keep the server alive: let go of the 169 GB)
    return {"n_requests": n_req, "views": views, "frame": "256x256 PNG", "prompt_lengths": len(texts), "warmup_requests": n_warm,
            "p50_ms": round(float(np.median(lat)), 2), "p90_ms": round(float(np.percentile(lat, 90)), 2),
            "max_ms": round(float(lat.max()), 2), "host_threads": srv.host_threads,
            "cgroup": {k: v - cg0.get(k, 0) for k, v in hostcpu.throttle_stats().items()},
            **({"all_ms": [round(float(x), 1) for x in lat], "stage_ms": {k: v[n_warm:] for k, v in (srv.stage_ms or {}).items()},
                "inner_ms": {k: v[-n_req:] for k, v in inner.items()}}
               if os.environ.get("DXA_BENCH_PF_DUMP") else {})}


def secondary_workloads(timeout_s: float = 150.0) -> dict:
    import subprocess
    out = {}
    for key, script, argv in (("db_pi0_finetune", "pi0_bench.py", ["3", "16"]), ("memvla_finetune", "memvla_bench.py", ["3", "16"]),
                              ("discrete_decode", "decode_bench.py", ["32", "--json"])):
        try:
            r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", script)] + argv, capture_output=True, text=True,
                               timeout=timeout_s)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            out[key] = json.loads(line[-1]) if line else {"error": (r.stderr or r.stdout)[-300:]}
        except Exception as e:  # noqa: BLE001  (never break the headline line)
            out[key] = {"error": repr(e)[:200]}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=16, help="episodes per GPU per step")
    ap.add_argument("--views", type=int, default=1)
    ap.add_argument("--s-text", dest="s_text", type=int, default=32)
    ap.add_argument("--dtype", default="bfloat16", choices=["bfloat16", "float32"])
    ap.add_argument("--llm-layers", dest="llm_layers", type=int, default=28)
    ap.add_argument("--vit-layers", dest="vit_layers", type=int, default=24)
    ap.add_argument("--force-reducer", action="store_true",
                    help="debug: run the RCCL gradient reducer even at world size 1 (exercises the DP code path)")
    ap.add_argument("--native-avg", dest="native_avg", action="store_true",
                    help="with --force-reducer at world size 1: the exact N > 1 collective sequence (in-place reduce_scatter(AVG) + "
                         "all_gather on the communication stream under the backward) — provokes the RCCL / GEMM CU contention on one GPU")
    ap.add_argument("--grad-comm", dest="grad_comm", default="bfloat16", choices=["bfloat16", "float32"],
                    help="dtype of the data-parallel gradient all-reduce (the reference's DeepSpeed bf16 run reduces bf16)")
    ap.add_argument("--grad-dtype", dest="grad_dtype", default="float32", choices=["bfloat16", "float32"],
                    help="dtype of the gradient arena the dW products write and AdamW reads: float32 (default: every gradient in "
                         "fp32); bfloat16 = the reference's DeepSpeed bf16 recipe (script/deepspeed/zero3.json: bf16 gradients, "
                         "fp32 masters in the optimizer) — measured 248.3 vs 249.6 ms/step, i.e. no real gain: inside the "
                         "power-limited step the dW products take the same time whatever they store")
    ap.add_argument("--static-batch", dest="static_batch", action="store_true",
                    help="round-1/2 behaviour: ONE device-resident batch re-used every step (splice plan served from the cache, no "
                         "uploads).  Default: 8 distinct host batches rotated, fresh token ids every step, images / actions uploaded "
                         "from pinned host memory on a copy stream one step ahead (dexbotic_amd/data/feeder.py)")
    ap.add_argument("--ragged", action="store_true", help="every fourth rotated batch has right-padded instructions")
    ap.add_argument("--overlap", dest="overlap", action="store_true",
                    help="AdamW segment by segment on a side stream under the next step's forward "
                         "(NativeTrainer(overlap_optimizer=True)).  Off by default: measured 247.8 vs 247.1 ms/step (round 3, "
                         "gpurun_out/r03_b_ov.json vs r03_b_noov.json) — the forward GEMMs slow from 254 to 363 us per launch while "
                         "the update streams beside them, which gives the hidden 36 ms back (DESIGN.md section 4)")
    ap.add_argument("--accum", type=int, default=1, help="gradient accumulation steps of the HEADLINE measurement")
    ap.add_argument("--recompute", action="store_true",
                    help="activation recompute (the reference's gradient_checkpointing=True, base_exp.py:245) for the headline "
                         "measurement: layers keep their input only and re-run their forward in backward.  Off: resident")
    ap.add_argument("--profile-stride", dest="profile_stride", type=int, default=5,
                    help="the live roofline times every n-th launch of each GEMM layout with HIP events (a deterministic sample over "
                         "the whole timed region; 1 = every launch: 1240 event pairs per step cost the step ~2 %%)")
    ap.add_argument("--no-recipe", dest="no_recipe", action="store_true",
                    help="skip the second figure: the reference recipe 8 episodes x 2 accumulation steps per GPU "
                         "(cogact_exp.py:41-46), same 16 episodes per optimizer step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the DB-pi0 / MemVLA secondary workloads (BASELINE.json configs[3], [4]; run as isolated "
                         "subprocesses after the headline measurement, single GPU only)")
    ap.add_argument("--cpu-threads", dest="cpu_threads", type=int, default=0,
                    help="cpu_baseline: use exactly this many threads (default: sweep {all, 1/2, 1/4, 16, 8} host cores on one "
                         "depth-1 step and keep the fastest)")
    ap.add_argument("--cpu-port", dest="cpu_port", action="store_true",
                    help="time the CPU oracle (port) even where /root/reference is importable")
    ap.add_argument("--no-latency", action="store_true")
    ap.add_argument("--no-shard", dest="no_shard", action="store_true",
                    help="N > 1: every rank repeats the full AdamW on all-gathered gradients (rounds 1-5) instead of the sharded "
                         "optimizer step (engine.ShardPlan: reduce-scatter -> own-shard update -> all-gather of the bf16 shadows)")
    ap.add_argument("--no-dp-emulation", dest="no_dp_emulation", action="store_true",
                    help="skip dp8_emulated_ms_per_step (N = 1 only: the LOCAL work of one rank of an 8-rank step — RCCL collectives over "
                         "the whole slices at world size 1 + AdamW over 1 / 8 of the arena; sharded and replicated, a few steps each)")
    args = ap.parse_args()

    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the native path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1 or args.force_reducer:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=device)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from dexbotic_amd import _lib as L
    from dexbotic_amd import kernels as K
    from dexbotic_amd.engine import OptimConfig
    from dexbotic_amd.trainer import NativeTrainer

    # torch's intra-op pool inside the container's CPU quota (dexbotic_amd/hostcpu.py: 128 OpenMP workers on a 16-CPU cgroup get the
    # whole process throttled; DXA_HOST_THREADS=0 leaves torch's default for an A/B)
    from dexbotic_amd import hostcpu
    host_threads = hostcpu.limit_host_threads()
    model, cfg, llm, vis = build_model(args, device)
    model.train()
    if args.recompute:
        model.gradient_checkpointing_enable()
    trainer = NativeTrainer(model, OptimConfig(base_lr=2e-5, weight_decay=0.0, max_grad_norm=1.0),
                            total_steps=1000, force_reducer=args.force_reducer,
                            grad_comm_dtype=getattr(torch, args.grad_comm), grad_accum=args.accum,
                            grad_dtype=getattr(torch, args.grad_dtype) if args.dtype == "bfloat16" else torch.float32,
                            overlap_optimizer=args.overlap, native_avg_world1=args.native_avg,
                            shard_optimizer=False if args.no_shard else None)
    if trainer.reducer is not None:
        trainer.reducer.time_comm = True
    from dexbotic_amd.data.feeder import DeviceFeeder
    micro_b = args.batch // args.accum
    if args.static_batch:
        fixed = synthetic_batch(micro_b, args.views, args.s_text, device, seed=1234 + rank)
        feed = iter(lambda: fixed, None)
    else:
        feed = DeviceFeeder(rotating_batches(host_batches(8, micro_b, args.views, args.s_text, 1234 + 100 * rank, args.ragged),
                                             seed=99 + rank), device)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def one_step():
        for _ in range(args.accum):
            loss_ = trainer.step(next(feed))
        return loss_

    for _ in range(args.warmup):
        one_step()
    sync()
    in_dt = L.BF16 if args.dtype == "bfloat16" else L.F32
    # dominant kernel: gemm_pp_kernel, whose three instantiations carry every large product of the step:
    # NT (forward linears, bf16 out), NN (dX = dY W, bf16 out), TN (dW = dY^T X, fp32 out into the gradient arena)
    prof_keys = {"NT fwd": (L.NT, in_dt, in_dt), "NN dX": (L.NN, in_dt, in_dt), "TN dW": (L.TN, in_dt, L.F32)}
    prof = K.GemmProfile(*set(prof_keys.values()), stride=args.profile_stride)
    K.GEMM_PROFILE = prof if rank == 0 else None
    cg0 = hostcpu.throttle_stats()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = one_step()
    sync()                                  # device-wide: also covers the optimizer update still in flight on its side stream
    dt = time.perf_counter() - t0
    cg1 = hostcpu.throttle_stats()
    K.GEMM_PROFILE = None
    comm_stats = None
    if trainer.reducer is not None and not trainer.reducer.local_only:
        # (before the recipe figure below runs more optimizer steps through the same reducer)
        red = trainer.reducer
        n_opt = max(trainer.global_step, 1)
        win = red.comm_window_ms()[-args.steps:]
        comm_stats = {"grad_comm_dtype": args.grad_comm, "grad_sync": red.algo,
                      "allreduce_gb_per_step": round(red.bytes_reduced / n_opt / 1e9, 3),
                      "collectives_per_step": round(red.collectives / n_opt, 1),
                      # first collective's start -> last collective's end on the communication stream (it runs under the
                      # backward: the part of it that is NOT hidden is what ms_per_step grows by against the 1-GPU line)
                      "comm_window_ms_per_step": round(float(np.mean(win)), 2) if win else None,
                      # the sharded optimizer step (default for N > 1): gradients reduce-scattered, AdamW over this rank's shard,
                      # updated bf16 shadows (+ the fp32 head's masters) all-gathered under the next forward
                      "optimizer_step": ("sharded: reduce-scatter -> adamw over 1/N -> all-gather of the updated weights"
                                         if trainer.sharded else "replicated: reduce-scatter + all-gather of gradients, full adamw per rank"),
                      "weights_gathered_gb_per_step": round(red.bytes_gathered / n_opt / 1e9, 3) if trainer.sharded else None}
    recipe = None
    if not args.no_recipe and args.accum == 1 and args.batch % 2 == 0 and not args.static_batch:
        # second figure: the reference recipe, 8 episodes x 2 accumulation steps per GPU per optimizer
        # step (cogact_exp.py:41-46), measured twice: pass by pass (half-height GEMM grids; the linears' dW is one product over both
        # micro-batches' (dY, X) pairs) and with the group coalesced into one pass
        trainer.set_grad_accum(2)
        feed2 = DeviceFeeder(rotating_batches(host_batches(8, args.batch // 2, args.views, args.s_text, 4321 + 100 * rank), seed=7 + rank),
                             device)
        n_rec = max(3, args.steps // 2)

        def recipe_leg():
            for _ in range(2 * 2):
                trainer.step(next(feed2))
            sync()
            tr0 = time.perf_counter()
            for _ in range(2 * n_rec):
                trainer.step(next(feed2))
            sync()
            rdt = torch.tensor([time.perf_counter() - tr0], device=device, dtype=torch.float64)
            if world > 1:
                dist.all_reduce(rdt, op=dist.ReduceOp.MAX)
            return {"episodes_per_s": round(args.batch * world * n_rec / float(rdt.item()), 3),
                    "ms_per_optimizer_step": round(1e3 * float(rdt.item()) / n_rec, 2)}
        two_pass = recipe_leg()                 # micro-batch by micro-batch, as HF's loop hands them over
        # ... and as exp/trainer.NativeDexboticTrainer runs that recipe by default: the two micro-batches of an optimizer step
        # held and run as ONE 16-episode pass (trainer.NativeTrainer coalesce_micro_batches; same mean loss, gradients equal
        # up to fp32 summation order: tests/test_hf_trainer_gpu.py)
        trainer.coalesce = bool(getattr(model, "coalescible_micro_batches", False))
        c0 = trainer.coalesced_steps
        one_pass = recipe_leg() if trainer.coalesce else None
        assert one_pass is None or trainer.coalesced_steps - c0 == n_rec + 2, "the coalesced leg did not coalesce"
        trainer.coalesce = False
        recipe = dict(one_pass or two_pass, optimizer_steps=n_rec, micro_batch=args.batch // 2, grad_accum=2,
                      execution=("the 2 micro-batches of an optimizer step coalesced into one pass (NativeDexboticTrainer default)"
                                 if one_pass else "micro-batch by micro-batch"),
                      two_passes=two_pass)
        trainer.set_grad_accum(1)
        del feed2
    tmax = torch.tensor([dt], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    loss_val = float(loss.item())
    peak_main = torch.cuda.max_memory_allocated(device)
    dp_emu = None
    if world == 1 and not args.no_dp_emulation and not args.force_reducer and args.dtype == "bfloat16" and args.accum == 1 \
            and not args.overlap:
        # what ONE rank of an 8-rank data-parallel step does locally, on the one GPU there is (SCALE runs need an 8-GPU node this
        # repo never had): the RCCL collectives of the real sequence over the whole slices at world size 1 (bf16 exchange, SUM) on
        # the communication stream under the backward, the sum of squares, AdamW — sharded: over rank 0's 1/8 of the arena, then
        # the all-gather of the updated bf16 shadows under the next forward; replicated: reduce-scatter + all-gather of the
        # gradients and the full AdamW (rounds 1-5).  NOT a scaling measurement: no xGMI transfer happens, the other ranks'
        # shards are simply not updated.  The main trainer's moments (64 GB) go first.
        try:
            ms_plain = 1e3 * dt / args.steps
            del trainer
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29517")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
            if not dist.is_initialized():
                dist.init_process_group("nccl", device_id=device)
            dp_emu = {"world_emulated": 8, "one_gpu_plain_ms_per_step": round(ms_plain, 2), "steps": max(4, args.steps // 2),
                      "what": "local work of one rank of 8: collectives at world size 1 over the whole slices (bf16 exchange) + "
                              "sum of squares + AdamW; no xGMI traffic — an estimate of the per-rank step, not a scaling measurement"}
            for key, kw in (("sharded", dict(shard_optimizer=True, emulate_world=8)), ("replicated", dict(shard_optimizer=False))):
                tr_e = NativeTrainer(model, OptimConfig(base_lr=2e-5, weight_decay=0.0, max_grad_norm=1.0), total_steps=1000,
                                     force_reducer=True, grad_comm_dtype=torch.bfloat16, **kw)
                for _ in range(2):
                    tr_e.step(next(feed))
                sync()
                te0 = time.perf_counter()
                for _ in range(dp_emu["steps"]):
                    tr_e.step(next(feed))
                sync()
                dp_emu[key + "_ms_per_step"] = round(1e3 * (time.perf_counter() - te0) / dp_emu["steps"], 2)
                if key == "sharded":
                    dp_emu["plan"] = tr_e.reducer.plan.describe()
                    dp_emu["weights_gathered_gb_per_step"] = round(tr_e.reducer.bytes_gathered / max(tr_e.global_step, 1) / 1e9, 3)
                    dp_emu["gradients_reduced_gb_per_step"] = round(tr_e.reducer.bytes_reduced / max(tr_e.global_step, 1) / 1e9, 3)
                tr_e.consolidate()
                model.store.on_bucket_ready = None
                del tr_e
                gc.collect()
                torch.cuda.empty_cache()
            trainer = None
        except Exception as e:  # noqa: BLE001  (never break the headline line)
            dp_emu = {"error": repr(e)[:300]}
            trainer = None

    f_fwd, S = flops_per_sample_fwd(llm.to_dict(), vis.to_dict(), args.views, args.s_text, 768, 12, 16, 4)
    samples = args.batch * world * args.steps
    value = samples / dt
    result = {
        "metric": "episodes/sec DB-CogACT fine-tune", "value": round(value, 3), "unit": "episodes/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 2),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if in_dt == L.BF16 else "f32",
        "data": "synthetic", "loss": round(loss_val, 5),
        "config": {"workload": f"DB-CogACT fine-tune step (CLIP-L/14@224 x{args.views} view, Qwen2.5-7B-class "
                               f"{args.llm_layers}L decoder, DiT-B head, AdamW), {args.batch} episodes/GPU, "
                               f"{args.s_text}-token instruction, S={S}",
                   "global_batch": args.batch * world, "seq_len": S, "parallelism": f"dp{world}",
                   "params_billion": round(model.store.total / 1e9, 3)},
        "tflops_per_gpu_model": round(value * 3 * f_fwd / world / 1e12, 1),
        "mfu_bf16": round(value * 3 * f_fwd / world / 1e12 / PEAK_BF16_TFLOPS, 4),
    }
    result["grad_dtype"] = "bf16" if model.store.bf16_grads else "f32"
    result["host"] = {"cpu_quota": hostcpu.cpu_quota(), "logical_cpus": os.cpu_count(), "torch_threads": host_threads,
                      "timed_region_cgroup": {k: cg1[k] - cg0.get(k, 0) for k in cg1}}
    if recipe is not None:
        result["reference_recipe_8x_accum2"] = recipe
    result["config"]["inputs"] = ("one device-resident batch re-used" if args.static_batch else
                                  "8 host batches rotated, fresh token ids every step, images/actions uploaded from pinned "
                                  "memory on a copy stream one step ahead" + (", every 4th batch right-padded" if args.ragged else ""))
    result["config"]["optimizer"] = "AdamW serial" if not args.overlap else "AdamW overlapped with the next forward (side stream, per-bucket events)"
    result["config"]["grad_accum"] = args.accum
    result["config"]["activations"] = "recomputed in backward (gradient checkpointing)" if args.recompute else "resident"
    result["peak_hbm_gb"] = round(peak_main / 1e9, 1)
    if comm_stats is not None:
        result.update(comm_stats)
    if dp_emu is not None:
        result["dp8_emulated_ms_per_step"] = dp_emu.get("sharded_ms_per_step")
        result["dp8_emulated"] = dp_emu
    if rank == 0:
        n, ms, fl, by = prof.summary()
        ach = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        # HBM bytes per launch of that kernel come from the committed rocprofv3 PMC passes over this same command
        # (separate --pmc runs; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950, checked on a known byte count)
        traffic = None
        pmc_path = next((pth for pth in (os.path.join(ROOT, "profiles", f"r0{r}_pmc.json") for r in (6, 5, 4, 3, 2)) if os.path.exists(pth)), None)
        pmc_commit = None
        if pmc_path and in_dt == L.BF16:
            with open(pmc_path) as f:
                pmc = json.load(f)
            traffic, pmc_commit = pmc.get("hbm_bytes_per_launch"), pmc.get("collected_on_commit")
        by_layout = {}
        for tag, key in prof_keys.items():
            kn, kms, kfl, kby = prof.summary(key)
            if kn:
                by_layout[tag] = {"launches": prof.launches(key), "timed": kn, "avg_launch_us": round(1e3 * kms / kn, 1),
                                  "achieved": round(kfl / (kms * 1e-3) / 1e12, 1)}
        result["roofline"] = {"bound": "mfma",
                              "kernel": "gemm_pp_kernel (dxa_gemm, 16-bit operands: the NT / NN / TN instantiations of the one "
                                        "256x256x64 ping-pong MFMA kernel = every forward linear, dX and dW product of the step)",
                              "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                              "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
                              "traffic_source": (os.path.relpath(pmc_path, ROOT) if pmc_path else "-") +
                                                " (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE per launch, separate "
                                                "passes, scripts/pmc_passes.sh).  FETCH_SIZE counts what leaves the "
                                                "XCD L2s, Infinity-Cache hits included: this is L2-miss (fabric-side) traffic, an upper "
                                                "bound on HBM bytes, not HBM bytes",
                              "traffic_kind": "l2_miss_bytes_per_launch",
                              # a committed measurement, not one of THIS run (PMC passes serialise the kernels): the commit whose
                              # gemm.hip it was collected on is carried so that a later kernel change shows as a mismatch
                              "traffic_collected_on_commit": pmc_commit,
                              "launches": prof.launches(), "launches_timed": n,
                              "timing": f"HIP events around every {prof.stride}-th launch of each layout, over the whole timed region"
                                        if prof.stride > 1 else "HIP events around every launch of the timed region",
                              "avg_launch_us": round(1e3 * ms / max(n, 1), 1),
                              "avg_launch_gflop": round(fl / max(n, 1) / 1e9, 2),
                              "algorithmic_bytes_per_launch": int(by / max(n, 1)), "by_layout": by_layout}
        if not args.no_latency and world == 1:
            model.eval()
            # BASELINE.json configs[1]: batch 1, 2 views.  SURVEY.md section 8(d): p50 over 200 requests — here 8 DIFFERENT
            # requests (token ids, both views) rotated, fresh sampler noise drawn inside every call
            reqs = [synthetic_batch(1, 2, args.s_text, device, seed=7 + 13 * i) for i in range(8)]
            norms = {"min": [-1.0] * 7, "max": [1.0] * 7}
            n_warm, n_req = 10, 200
            lat = []
            for i in range(n_warm + n_req):
                b1 = reqs[i % len(reqs)]
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                model.inference_action(b1["input_ids"], b1["images"], {"cfg_scale": 1.5, "num_ddim_steps": 10,
                                                                        "action_norms": norms})
                lat.append(1e3 * (time.perf_counter() - t1))       # inference_action ends with a .cpu() sync
            lat = np.asarray(lat[n_warm:])
            result["p50_action_inference_ms"] = round(float(np.median(lat)), 2)
            result["action_inference"] = {"n_requests": n_req, "distinct_inputs": len(reqs), "p50_ms": round(float(np.median(lat)), 2),
                                          "p90_ms": round(float(np.percentile(lat, 90)), 2), "min_ms": round(float(lat.min()), 2)}
            # roofline of the request: algorithmic FLOPs of ONE two-view request (ViT x 2 views +
            # projector + 28 decoder layers at S = 543 + 10 DDIM steps x CFG pair of DiT-B forwards) over the p50 latency against the
            # dense bf16 MFMA peak, and the bf16 weight bytes a request must touch at least once over the same time against HBM
            f_req, s_req = flops_per_sample_fwd(llm.to_dict(), vis.to_dict(), 2, args.s_text, 768, 12, 16, 0)
            f_dit = 10 * 2 * (12 * (2 * 12 * 768 * 768 * 17 + 4 * 17 * 17 * 768) + 2 * (3584 * 768 + 256 * 768 + 768 * 768))
            lw = llm.to_dict()
            dl, fl_, hq, hkv = lw["hidden_size"], lw["intermediate_size"], lw["num_attention_heads"], lw["num_key_value_heads"]
            w_llm = lw["num_hidden_layers"] * (dl * (hq + 2 * hkv) * (dl // hq) + dl * dl + 3 * dl * fl_)
            vw = vis.to_dict()
            w_vit = (vw["num_hidden_layers"] - 1) * (4 * vw["hidden_size"] ** 2 + 2 * vw["hidden_size"] * vw["intermediate_size"])
            w_bytes = 2.0 * (w_llm + w_vit + vw["hidden_size"] * dl + dl * dl) + 2.0 * 12 * 12 * 768 * 768
            p50_s = float(np.median(lat)) * 1e-3
            result["inference_roofline"] = {
                "bound": "mfma", "achieved": round((f_req + f_dit) / p50_s / 1e12, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                "frac": round((f_req + f_dit) / p50_s / 1e12 / PEAK_BF16_TFLOPS, 4),
                "request_tflop": round((f_req + f_dit) / 1e12, 3), "seq_len": s_req,
                "weight_stream_tb_s": round(w_bytes / p50_s / 1e12, 3), "weight_gb": round(w_bytes / 1e9, 2),
                "hbm_frac_of_8_tb_s": round(w_bytes / p50_s / 8e12, 4),
                "what": "whole request (host launch of the replayed graph + device + result copy) at its p50, not one kernel: the "
                        "per-kernel table is profiles/r06_infer_kernel_stats.txt"}
            result["config"]["inference_workload"] = ("DB-CogACT bf16 action inference, batch 1, 2 views 224x224, "
                                                      "32-token instruction (S=543), CFG 1.5, 10 DDIM steps, through inference_action (HIP-graph replay from the third request of a shape on)")
            try:
                result["action_inference"]["process_frame"] = process_frame_latency(model, n_req=50)
            except Exception as e:  # noqa: BLE001  (never break the headline line)
                result["action_inference"]["process_frame"] = {"error": repr(e)[:200]}
        if not args.no_cpu_baseline and world == 1:
            try:
                result["cpu_baseline"] = cpu_baseline(args, llm, vis)
            except Exception as e:  # noqa: BLE001  (the baseline must never break the bench line)
                result["cpu_baseline"] = {"value": None, "error": repr(e)[:200]}
        if not args.no_secondary and world == 1:
            # BASELINE.json configs[3] / [4] as driver-visible lines: each runs in its own process (its own arenas, a crash
            # or a timeout there cannot take the headline line with it) once this process has given the GPU memory back
            del trainer, model, feed
            reqs = b1 = None
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            result["secondary"] = secondary_workloads()
        print(json.dumps(result), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
