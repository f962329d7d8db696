"""Generate tests/golden/*.npz by running the REFERENCE's own Python classes on CPU.

TEST INFRASTRUCTURE.  Runs only in the build container (needs /root/reference); the produced
fixtures are committed so nothing at test/bench time reads /root/reference.

    python -m oracle.gen_golden            # from the repo root

What is pinned (SURVEY.md §8c "oracle outputs the new repo should freeze"):
  * state_dict key map + shapes of CogACTForCausalLM (asserted == oracle.weights.cogact_shapes)
  * vision features, spliced inputs_embeds/attention_mask, final hidden, cognition feature,
    x_t, eps_hat, loss for INJECTED (noise, t, CFG-drop) draws
  * selected gradients + global grad-norm, parameters after one clip(1.0)+AdamW step, 2nd-step loss
  * DDIM(10, cfg 1.5) trajectory + de-normalised [16,7] chunk for injected initial noise
  * action normalise / bin / string / de-normalise integer rows (round-half-even)
Shims are the three documented in SURVEY.md §8(c): timm Attention/Mlp restatement installed after
`import transformers`; a locally saved tiny CLIP directory; nothing else for the CogACT path.
"""
from __future__ import annotations

import importlib.machinery
import os
import sys
import tempfile
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")


def install_timm_shim():
    """timm is an un-vendored, unpinned dependency (pyproject.toml:33) absent from this image.
    Restated semantics of timm.models.vision_transformer.{Attention,Mlp} as the reference uses
    them (dit.py:11,145-157)."""
    import transformers  # noqa: F401  (must be imported first, SURVEY §8c shim 1)

    class Attention(nn.Module):
        def __init__(self, dim, num_heads=8, qkv_bias=False, **kw):
            super().__init__()
            self.num_heads = num_heads
            self.head_dim = dim // num_heads
            self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
            self.proj = nn.Linear(dim, dim)

        def forward(self, x):
            B, N, C = x.shape
            qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, self.head_dim).permute(2, 0, 3, 1, 4)
            q, k, v = qkv.unbind(0)
            x = F.scaled_dot_product_attention(q, k, v)
            return self.proj(x.transpose(1, 2).reshape(B, N, C))

    class Mlp(nn.Module):
        def __init__(self, in_features, hidden_features=None, out_features=None,
                     act_layer=nn.GELU, drop=0.0, **kw):
            super().__init__()
            self.fc1 = nn.Linear(in_features, hidden_features)
            self.act = act_layer()
            self.fc2 = nn.Linear(hidden_features, out_features or in_features)

        def forward(self, x):
            return self.fc2(self.act(self.fc1(x)))

    timm = types.ModuleType("timm")
    timm.models = types.ModuleType("timm.models")
    vt = types.ModuleType("timm.models.vision_transformer")
    vt.Attention, vt.Mlp = Attention, Mlp
    timm.models.vision_transformer = vt
    for n, m in (("timm", timm), ("timm.models", timm.models), ("timm.models.vision_transformer", vt)):
        m.__spec__ = importlib.machinery.ModuleSpec(n, None)
        sys.modules[n] = m


def build_reference(cfg, weights):
    """Instantiate the reference CogACTForCausalLM at OracleConfig `cfg` and load `weights`."""
    from transformers import CLIPImageProcessor, CLIPVisionConfig, CLIPVisionModel, Qwen2Config
    from dexbotic.model.cogact.action_model import action_models
    from dexbotic.model.cogact.action_model.dit import DiT
    from dexbotic.model.cogact.cogact_arch import CogActConfig, CogACTForCausalLM

    d = os.path.join(tempfile.mkdtemp(), "tiny_clip")
    vcfg = CLIPVisionConfig(hidden_size=cfg.v_hidden, intermediate_size=cfg.v_inter,
                            num_hidden_layers=cfg.v_layers, num_attention_heads=cfg.v_heads,
                            image_size=cfg.v_image, patch_size=cfg.v_patch, layer_norm_eps=cfg.v_eps)
    CLIPVisionModel(vcfg).save_pretrained(d)
    CLIPImageProcessor(size={"shortest_edge": cfg.v_image},
                       crop_size={"height": cfg.v_image, "width": cfg.v_image}).save_pretrained(d)
    # register a small DiT size with the reference's own class (registry: action_models.py:60)
    action_models.DiT_models["DiT-T"] = lambda **kw: DiT(
        depth=cfg.dit_depth, hidden_size=cfg.dit_hidden, num_heads=cfg.dit_heads, **kw)
    llm = Qwen2Config(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size,
                      intermediate_size=cfg.intermediate_size, num_hidden_layers=cfg.num_hidden_layers,
                      num_attention_heads=cfg.num_attention_heads,
                      num_key_value_heads=cfg.num_key_value_heads, max_position_embeddings=4096,
                      rope_theta=cfg.rope_theta, rms_norm_eps=cfg.rms_norm_eps)
    c = CogActConfig(llm_config=llm, mm_vision_tower=d, mm_projector_type="mlp2x_gelu",
                     action_model_type="DiT-T", action_dim=cfg.action_dim, chunk_size=cfg.chunk_size)
    m = CogACTForCausalLM(c)
    ref_shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    from oracle.weights import cogact_shapes
    mine = cogact_shapes(cfg)
    assert ref_shapes == mine, (set(ref_shapes) ^ set(mine),
                                {k: (ref_shapes[k], mine[k]) for k in ref_shapes
                                 if k in mine and ref_shapes[k] != mine[k]})
    if weights is not None:                      # None: keep the reference's own random initialisation (bench.py timing)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()}, strict=True)
    # CogACTModelConfig._freeze_model (cogact_exp.py:106-124) with the default freeze_* = False:
    # every parameter under model.model trains (CLIPVisionTower.load_model froze the tower).
    for p_ in m.model.parameters():
        p_.requires_grad = True
    return m


class inject_rng:
    """Replace the global-RNG draws inside ActionModel.loss / LabelEmbedder.token_drop /
    inference_action (action_models.py:106-109, dit.py:85-87, cogact_arch.py:163-168) by given
    tensors so reference and replacement see identical randomness."""

    def __init__(self, noise=None, timesteps=None, drop_u=None, init_noise=None):
        self.noise, self.timesteps, self.drop_u, self.init_noise = noise, timesteps, drop_u, init_noise

    def __enter__(self):
        self._o = (torch.randn_like, torch.randint, torch.rand, torch.randn)
        o_randn_like, o_randint, o_rand, o_randn = self._o

        def randn_like(x, *a, **k):
            if self.noise is not None and tuple(x.shape) == tuple(self.noise.shape):
                return self.noise.to(x.dtype).clone()
            return o_randn_like(x, *a, **k)

        def randint(*a, **k):
            if self.timesteps is not None:
                return self.timesteps.clone()
            return o_randint(*a, **k)

        def rand(*a, **k):
            if self.drop_u is not None:
                return self.drop_u.clone()
            return o_rand(*a, **k)

        def randn(*a, **k):
            if self.init_noise is not None:
                return self.init_noise.to(k.get("dtype", torch.float32)).clone()
            return o_randn(*a, **k)

        torch.randn_like, torch.randint, torch.rand, torch.randn = randn_like, randint, rand, randn
        return self

    def __exit__(self, *exc):
        torch.randn_like, torch.randint, torch.rand, torch.randn = self._o


GRAD_KEYS = [
    "model.llm.layers.0.self_attn.q_proj.bias",
    "model.llm.layers.0.self_attn.k_proj.bias",
    "model.llm.layers.0.input_layernorm.weight",
    "model.llm.layers.1.post_attention_layernorm.weight",
    "model.llm.norm.weight",
    "model.mm_projector.0.bias",
    "model.mm_projector.2.bias",
    "model.mm_vision_tower.vision_tower.embeddings.class_embedding",
    "model.mm_vision_tower.vision_tower.pre_layrnorm.weight",
    "model.mm_vision_tower.vision_tower.encoder.layers.0.self_attn.q_proj.bias",
    "model.mm_vision_tower.vision_tower.encoder.layers.1.mlp.fc1.bias",
    "model.action_head.net.final_layer.linear.weight",
    "model.action_head.net.blocks.0.attn.qkv.bias",
    "model.action_head.net.t_embedder.mlp.0.bias",
    "model.action_head.net.z_embedder.linear.bias",
    "model.action_head.net.x_embedder.linear.weight",
    "model.action_head.net.positional_embedding",
]
# big matrices: store a strided sample + the Frobenius norm
GRAD_SAMPLED = [
    "model.llm.layers.0.self_attn.q_proj.weight",
    "model.llm.layers.0.self_attn.v_proj.weight",
    "model.llm.layers.1.mlp.down_proj.weight",
    "model.llm.layers.0.mlp.gate_proj.weight",
    "model.llm.embed_tokens.weight",
    "model.mm_projector.0.weight",
    "model.mm_vision_tower.vision_tower.embeddings.patch_embedding.weight",
    "model.mm_vision_tower.vision_tower.embeddings.position_embedding.weight",
    "model.mm_vision_tower.vision_tower.encoder.layers.0.mlp.fc2.weight",
    "model.action_head.net.blocks.1.mlp.fc1.weight",
    "model.action_head.net.z_embedder.linear.weight",
]


def no_decay_name(n: str) -> bool:
    """Restated grouping rule of OptimizerConfig._get_optimizer_grouped_parameters
    (base_exp.py:95-203) under the pinned transformers 4.51: decay = every parameter that is not
    inside an nn.LayerNorm module and has no "bias" in its name."""
    if "bias" in n:
        return True
    ln = ("layer_norm1.", "layer_norm2.", "pre_layrnorm.", "post_layernorm.")
    return any(t in n for t in ln)


def gen_cogact(tag: str, cfg, seed: int, B: int, L: int, lengths, views: int):
    from oracle.weights import cogact_shapes, make_weights, weights_crc
    w = make_weights(cogact_shapes(cfg), seed)
    m = build_reference(cfg, w)
    m.train()
    rs = np.random.RandomState(seed + 1)
    ids = rs.randint(10, cfg.vocab_size - 10, size=(B, L)).astype(np.int64)
    ids[:, 1] = -200
    mask = np.zeros((B, L), dtype=bool)
    for b, n in enumerate(lengths):
        mask[b, :n] = True
    img_shape = (B, views, 3, cfg.v_image, cfg.v_image) if views > 1 else (B, 3, cfg.v_image, cfg.v_image)
    images = np.clip(rs.standard_normal(img_shape), -2.5, 2.5).astype(np.float32)
    actions = rs.uniform(-1, 1, size=(B, cfg.chunk_size * cfg.action_dim)).astype(np.float32)
    R = 4
    noise = rs.standard_normal((R * B, cfg.chunk_size, cfg.action_dim)).astype(np.float32)
    timesteps = rs.randint(0, cfg.diffusion_steps, size=(R * B,)).astype(np.int64)
    drop_u = rs.uniform(0, 1, size=(R * B,)).astype(np.float32)
    drop_u[1] = 0.01                                   # force at least one dropped condition
    t = torch.from_numpy

    # capture intermediates with forward hooks on the reference modules
    cap = {}
    def _mk(name):
        def hook(mod, i, o):
            cap[name] = o.detach()
        return hook
    h1 = m.model.mm_projector.register_forward_hook(_mk("proj_out"))
    h2 = m.model.mm_vision_tower.register_forward_hook(_mk("vit_out"))
    orig_llm_fwd = m.model.llm.forward

    def llm_fwd(*a, **k):
        cap["inputs_embeds"] = k["inputs_embeds"].detach()
        cap["attention_mask"] = None if k.get("attention_mask") is None else k["attention_mask"].detach()
        return orig_llm_fwd(*a, **k)
    m.model.llm.forward = llm_fwd
    net = m.model.action_head.net
    def _net_hook(mod, i, o):
        cap["x_t"], cap["eps_hat"] = i[0].detach(), o.detach()
    h3 = net.register_forward_hook(_net_hook)

    def step():
        with inject_rng(noise=t(noise), timesteps=t(timesteps), drop_u=t(drop_u)):
            return m(input_ids=t(ids), attention_mask=t(mask), images=t(images), actions=t(actions),
                     labels=t(ids).clone())
    out = step()
    loss = out.loss
    loss.backward()
    res = dict(
        seed=np.int64(seed), weights_crc=np.int64(weights_crc(w)),
        input_ids=ids, attention_mask=mask, images=images, actions=actions,
        noise=noise, timesteps=timesteps, drop_u=drop_u,
        vit_out=cap["vit_out"].numpy(), proj_out=cap["proj_out"].numpy(),
        inputs_embeds=cap["inputs_embeds"].numpy(), new_attention_mask=cap["attention_mask"].numpy(),
        logits=out.logits.detach().numpy(), x_t=cap["x_t"].numpy(), eps_hat=cap["eps_hat"].numpy(),
        loss=np.float64(loss.item()),
    )
    grads = {n: p.grad for n, p in m.named_parameters()}
    for k in GRAD_KEYS:
        res["grad/" + k] = grads[k].numpy().copy()
    for k in GRAD_SAMPLED:
        g = grads[k].reshape(-1)
        res["gradS/" + k] = g[::97].numpy().copy()
        res["gradN/" + k] = np.float64(g.double().norm().item())
    never = sorted(n for n, g in grads.items() if g is None)
    res["no_grad_params"] = np.array(never)
    total_norm = torch.nn.utils.clip_grad_norm_([p for p in m.parameters() if p.grad is not None], 1.0)
    res["grad_norm"] = np.float64(float(total_norm))
    # one AdamW step: 2 groups (decay / no-decay) as base_exp.py builds them; wd 0.01 to pin grouping
    named = [(n, p) for n, p in m.named_parameters() if p.grad is not None]
    groups = [dict(params=[p for n, p in named if not no_decay_name(n)], weight_decay=0.01),
              dict(params=[p for n, p in named if no_decay_name(n)], weight_decay=0.0)]
    opt = torch.optim.AdamW(groups, lr=1e-3, betas=(0.9, 0.999), eps=1e-8)
    opt.step()
    sd2 = m.state_dict()
    for k in GRAD_KEYS:
        res["param1/" + k] = sd2[k].numpy().copy()
    for k in GRAD_SAMPLED:
        res["param1S/" + k] = sd2[k].reshape(-1)[::97].numpy().copy()
    opt.zero_grad()
    res["loss_step2"] = np.float64(step().loss.item())
    h1.remove(); h2.remove(); h3.remove()
    m.model.llm.forward = orig_llm_fwd

    # inference on sample 0 with the ORIGINAL weights
    m.load_state_dict({k: t(v) for k, v in w.items()}, strict=True)
    m.eval()
    init_noise = rs.standard_normal((1, cfg.chunk_size, cfg.action_dim)).astype(np.float32)
    norms = {"min": (-1 - 0.1 * np.arange(cfg.action_dim)).tolist(),
             "max": (1 + 0.05 * np.arange(cfg.action_dim)).tolist()}
    n_tok = int(lengths[0])
    traj = []
    orig_ddim = None
    from dexbotic.model.cogact.action_model import diffusion as D
    orig_ddim = D.GaussianDiffusion.ddim_sample

    def rec(self, *a, **k):
        o = orig_ddim(self, *a, **k)
        traj.append(o["sample"].detach().numpy().copy())
        return o
    D.GaussianDiffusion.ddim_sample = rec
    try:
        with inject_rng(init_noise=t(init_noise)):
            acts = m.inference_action(t(ids[:1, :n_tok]), t(images[:1]),
                                      {"cfg_scale": 1.5, "num_ddim_steps": 10, "action_norms": norms})
    finally:
        D.GaussianDiffusion.ddim_sample = orig_ddim
    res.update(init_noise=init_noise, norm_min=np.array(norms["min"]), norm_max=np.array(norms["max"]),
               infer_ids=ids[:1, :n_tok], ddim_traj=np.stack(traj), infer_actions=np.array(acts, dtype=np.float64))
    os.makedirs(GOLD, exist_ok=True)
    np.savez_compressed(os.path.join(GOLD, f"cogact_{tag}.npz"), **res)
    print(tag, "loss", res["loss"], "grad_norm", res["grad_norm"], "loss2", res["loss_step2"],
          "no-grad params:", len(never))


def gen_lm(tag: str, cfg, seed: int, B: int, L: int, lengths, n_new: int):
    """Row A10: DexboticForCausalLM.forward with labels (lm_head + HF causal-LM cross-entropy, dexbotic_arch.py:
    429-496) and a greedy continuation.  generate() itself does not run under this container's transformers
    (SURVEY.md §8c shim iii), so the greedy ids come from a full-prefix recompute loop over the reference's own
    forward — the token sequence a KV-cached decode has to reproduce exactly."""
    from transformers import CLIPImageProcessor, CLIPVisionConfig, CLIPVisionModel, Qwen2Config
    from dexbotic.model.dexbotic_arch import DexboticConfig, DexboticForCausalLM
    from oracle.weights import cogact_shapes, make_weights, weights_crc
    w = {k: v for k, v in make_weights(cogact_shapes(cfg), seed).items() if ".action_head." not in k}
    d = os.path.join(tempfile.mkdtemp(), "tiny_clip")
    vcfg = CLIPVisionConfig(hidden_size=cfg.v_hidden, intermediate_size=cfg.v_inter, num_hidden_layers=cfg.v_layers,
                            num_attention_heads=cfg.v_heads, image_size=cfg.v_image, patch_size=cfg.v_patch,
                            layer_norm_eps=cfg.v_eps)
    CLIPVisionModel(vcfg).save_pretrained(d)
    CLIPImageProcessor(size={"shortest_edge": cfg.v_image},
                       crop_size={"height": cfg.v_image, "width": cfg.v_image}).save_pretrained(d)
    llm = Qwen2Config(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                      num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                      num_key_value_heads=cfg.num_key_value_heads, max_position_embeddings=4096,
                      rope_theta=cfg.rope_theta, rms_norm_eps=cfg.rms_norm_eps)
    m = DexboticForCausalLM(DexboticConfig(llm_config=llm, mm_vision_tower=d, mm_projector_type="mlp2x_gelu"))
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == {k: v.shape for k, v in w.items()}
    m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=True)
    for p_ in m.parameters():
        p_.requires_grad = True
    m.train()
    rs = np.random.RandomState(seed + 7)
    ids = rs.randint(10, cfg.vocab_size - 10, size=(B, L)).astype(np.int64)
    ids[:, 1] = -200
    mask = np.zeros((B, L), dtype=bool)
    for b, n in enumerate(lengths):
        mask[b, :n] = True
    labels = ids.copy()
    labels[:, :4] = -100                                   # prompt part is not supervised (like the SFT collator)
    labels[~mask] = -100
    images = np.clip(rs.standard_normal((B, 3, cfg.v_image, cfg.v_image)), -2.5, 2.5).astype(np.float32)
    t = torch.from_numpy
    out = m(input_ids=t(ids), attention_mask=t(mask), labels=t(labels), images=t(images))
    out.loss.backward()
    sd = dict(m.named_parameters())
    keep = ["lm_head.weight", "model.llm.norm.weight", "model.llm.layers.0.self_attn.q_proj.weight",
            "model.llm.layers.0.self_attn.k_proj.bias", "model.mm_projector.2.weight"]
    res = dict(weights_crc=np.uint32(weights_crc(w)), seed=np.int64(seed), input_ids=ids, attention_mask=mask,
               labels=labels, images=images, loss=np.float32(out.loss.item()),
               logits=out.logits.detach().numpy().astype(np.float32))
    gsq = 0.0
    for n, p_ in sd.items():
        if p_.grad is not None:
            gsq += float(p_.grad.double().pow(2).sum())
            res["gradN/" + n] = np.float64(p_.grad.double().norm().item())
    res["grad_norm"] = np.float64(gsq ** 0.5)
    for n in keep:
        res["grad/" + n] = sd[n].grad.numpy().astype(np.float32)
    emb_g = sd["model.llm.embed_tokens.weight"].grad
    rows = np.unique(ids[ids >= 0])[:6]
    res["embed_rows"] = rows
    res["grad_embed_rows"] = emb_g[t(rows)].numpy().astype(np.float32)
    # greedy continuation (batch 1, no padding) by full-prefix recompute through the reference forward
    m.eval()
    cur = t(ids[:1, :lengths[0]]).clone()
    img1 = t(images[:1])
    new, rows_l = [], []
    with torch.no_grad():
        for _ in range(n_new):
            lg = m(input_ids=cur, images=img1).logits[0, -1].float()
            nxt = int(torch.argmax(lg))
            new.append(nxt)
            rows_l.append(lg.numpy().astype(np.float32))
            cur = torch.cat([cur, torch.tensor([[nxt]], dtype=cur.dtype)], dim=1)
    res["decode_prompt"] = ids[:1, :lengths[0]]
    res["decode_new_ids"] = np.array(new, dtype=np.int64)
    res["decode_logits"] = np.stack(rows_l)
    top2 = np.sort(res["decode_logits"], axis=1)[:, -2:]
    res["decode_margin"] = (top2[:, 1] - top2[:, 0]).astype(np.float32)   # argmax gap: how decisive each choice is
    np.savez_compressed(os.path.join(GOLD, f"lm_{tag}.npz"), **res)
    print(f"[gen_golden] lm_{tag}: loss {res['loss']:.5f} |g| {res['grad_norm']:.4f} new ids {new} "
          f"min margin {res['decode_margin'].min():.4g}")


def gen_hybrid(tag: str, cfg, seed: int, B: int, L: int, lengths):
    """HybridCogACTForCausalLM.forward (hybrid_cogact_arch.py:59-207): text CE on the has_text samples + has_action-
    weighted diffusion loss, one backward.  Same tiny configuration / weights as cogact_<tag>."""
    from oracle.weights import cogact_shapes, make_weights, weights_crc
    from dexbotic.model.cogact import hybrid_cogact_arch as H
    w = make_weights(cogact_shapes(cfg), seed)
    base = build_reference(cfg, w)                      # CogACTForCausalLM: same config object and key map
    m = H.HybridCogACTForCausalLM(base.config)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=True)
    for p_ in m.parameters():
        p_.requires_grad = True
    m.train()
    rs = np.random.RandomState(seed + 11)
    ids = rs.randint(10, cfg.vocab_size - 10, size=(B, L)).astype(np.int64)
    ids[:, 1] = -200
    mask = np.zeros((B, L), dtype=bool)
    for b, n in enumerate(lengths):
        mask[b, :n] = True
    has_text = np.array([1, 0, 1][:B], dtype=np.int64)
    has_action = np.array([[1], [1], [0]][:B], dtype=np.int64)
    labels = ids.copy()
    labels[:, :5] = -100
    labels[~mask] = -100
    labels[has_text == 0] = -100                         # the data pipeline leaves no text targets on action-only samples
    images = np.clip(rs.standard_normal((B, 3, cfg.v_image, cfg.v_image)), -2.5, 2.5).astype(np.float32)
    actions = rs.uniform(-1, 1, size=(B, cfg.chunk_size * cfg.action_dim)).astype(np.float32)
    R = 4
    noise = rs.standard_normal((R * B, cfg.chunk_size, cfg.action_dim)).astype(np.float32)
    timesteps = rs.randint(0, cfg.diffusion_steps, size=(R * B,)).astype(np.int64)
    drop_u = rs.uniform(0, 1, size=(R * B,)).astype(np.float32)
    drop_u[2] = 0.01
    t = torch.from_numpy
    with inject_rng(noise=t(noise), timesteps=t(timesteps), drop_u=t(drop_u)):
        out = m(input_ids=t(ids), attention_mask=t(mask), labels=t(labels), images=t(images), actions=t(actions),
                has_action=t(has_action), has_text=t(has_text))
    out.loss.backward()
    sd = dict(m.named_parameters())
    res = dict(weights_crc=np.uint32(weights_crc(w)), seed=np.int64(seed), input_ids=ids, attention_mask=mask, labels=labels,
               images=images, actions=actions, has_action=has_action, has_text=has_text, noise=noise, timesteps=timesteps,
               drop_u=drop_u, loss=np.float32(out.loss.item()), text_loss=np.float32(out.text_loss.item()),
               action_loss=np.float32(out.action_loss.item()))
    gsq = 0.0
    for n, p_ in sd.items():
        if p_.grad is not None:
            gsq += float(p_.grad.double().pow(2).sum())
            res["gradN/" + n] = np.float64(p_.grad.double().norm().item())
    res["grad_norm"] = np.float64(gsq ** 0.5)
    for n in ("lm_head.weight", "model.action_head.net.final_layer.linear.weight", "model.llm.layers.0.self_attn.q_proj.weight",
              "model.mm_projector.2.weight"):
        res["grad/" + n] = sd[n].grad.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(GOLD, f"hybrid_{tag}.npz"), **res)
    print(f"[gen_golden] hybrid_{tag}: loss {res['loss']:.5f} = text {res['text_loss']:.5f} + action {res['action_loss']:.5f}"
          f" |g| {res['grad_norm']:.4f}")


def gen_action_bins():
    """Integer rows (A9): run the reference transform + _denorm + discrete decode."""
    from dexbotic.data.dataset.transform.action import ActionNormAnd2String
    from dexbotic.model.dexbotic_arch import ActionOutputForCausalLM
    from dexbotic.model.discrete_vla.discrete_vla_arch import DiscreteVLAForCausalLM
    rs = np.random.RandomState(7)
    A, T, V = 7, 64, 255
    mn = rs.uniform(-2, -0.5, size=A)
    mx = rs.uniform(0.5, 2, size=A)
    act = rs.uniform(-2.5, 2.5, size=(T, A))
    # exact half-way points: normalised value giving k+0.5 bins (round-half-even cases)
    halves = (np.arange(0, 40) + 0.5) / (V - 1) * 2 - 1
    tr = ActionNormAnd2String.__new__(ActionNormAnd2String)
    normed = tr._norm_action(act, mn, mx)
    normed_all = np.concatenate([normed.reshape(-1), halves, np.array([-1.0, 1.0, 0.0, -1.5, 1.5])])
    normed_all = normed_all[: (len(normed_all) // A) * A].reshape(-1, A)
    bins = tr._action2bin(normed_all, V)
    strs = tr._bin2string(bins, " {value}")

    class _D(ActionOutputForCausalLM):
        def inference_action(self, *a, **k):
            pass
    den = _D()._denorm(normed_all, {"min": mn.tolist(), "max": mx.tolist()})
    back = np.concatenate([DiscreteVLAForCausalLM._discrete_action_to_continuous(None, s, V) for s in strs])
    np.savez_compressed(os.path.join(GOLD, "action_bins.npz"), action=act, mn=mn, mx=mx, normed=normed,
                        normed_all=normed_all, bins=bins, strings=np.array(strs), denorm=den,
                        decoded=back, vocab=np.int64(V))
    print("action_bins", bins.shape, strs[0])


def gen_diffusion_tables():
    from dexbotic.model.cogact.action_model.diffusion import create_diffusion
    tr = create_diffusion("", "squaredcos_cap_v2", diffusion_steps=100, sigma_small=True, learn_sigma=False)
    out = dict(betas=tr.betas, alphas_cumprod=tr.alphas_cumprod,
               sqrt_alphas_cumprod=tr.sqrt_alphas_cumprod,
               sqrt_one_minus_alphas_cumprod=tr.sqrt_one_minus_alphas_cumprod)
    for n in (1, 2, 5, 10, 20, 25, 50):
        dd = create_diffusion(f"ddim{n}", "squaredcos_cap_v2", diffusion_steps=100, sigma_small=True,
                              learn_sigma=False)
        out[f"ddim{n}/timestep_map"] = np.array(dd.timestep_map)
        out[f"ddim{n}/alphas_cumprod"] = dd.alphas_cumprod
        out[f"ddim{n}/alphas_cumprod_prev"] = dd.alphas_cumprod_prev
        out[f"ddim{n}/sqrt_recip_alphas_cumprod"] = dd.sqrt_recip_alphas_cumprod
        out[f"ddim{n}/sqrt_recipm1_alphas_cumprod"] = dd.sqrt_recipm1_alphas_cumprod
    np.savez_compressed(os.path.join(GOLD, "diffusion_tables.npz"), **out)
    print("diffusion tables ok")


def main():
    sys.path.insert(0, REF)
    sys.path.insert(0, ROOT)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    install_timm_shim()
    from oracle.cogact_oracle import OracleConfig
    gen_diffusion_tables()
    gen_action_bins()
    # t1: one view, right padding on sample 1, GQA 2:1, hd 128 (LLM) / 64 (ViT, DiT)
    gen_cogact("t1", OracleConfig(), seed=1234, B=3, L=12, lengths=[12, 9, 11], views=1)
    # t2: two views, GQA 4:2, deeper
    cfg2 = OracleConfig(vocab_size=640, hidden_size=512, intermediate_size=768, num_hidden_layers=3,
                        num_attention_heads=4, num_key_value_heads=2, v_hidden=192, v_inter=384,
                        v_layers=4, v_heads=3, dit_hidden=192, dit_depth=3, dit_heads=3)
    gen_cogact("t2", cfg2, seed=4321, B=2, L=16, lengths=[16, 13], views=2)
    gen_lm("t1", OracleConfig(), seed=1234, B=3, L=12, lengths=[12, 9, 11], n_new=6)
    gen_hybrid("t1", OracleConfig(), seed=1234, B=3, L=12, lengths=[12, 9, 11])


if __name__ == "__main__":
    main()
