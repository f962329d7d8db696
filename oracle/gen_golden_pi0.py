"""Generate tests/golden/pi0_t1.npz by running the REFERENCE's Pi0ForCausalLM on CPU — TEST INFRASTRUCTURE.

    python -m oracle.gen_golden_pi0        # from the repo root, build container only (needs /root/reference)

Shims: timm stub after `import transformers`; a locally saved SigLIP processor directory; under this
container's transformers 5.x (i) GemmaTextScaledWordEmbedding.embed_scale is set to 1 so that the reference's own
`* sqrt(hidden)` (pi0_arch.py:247-250) is the only scale, as under the pinned 4.51, and (ii) DynamicCache gets
`key_cache` / `value_cache` views (pi0_arch.py:178-183 reads them).  Random draws of forward()/inference_action()
(torch.normal, Beta.sample) are replaced by injected tensors.
"""
from __future__ import annotations

import os
import sys
import tempfile

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")


def build_reference(cfg, weights):
    from transformers import DynamicCache, SiglipImageProcessor
    from dexbotic.model.pi0.pi0_arch import Pi0Config, Pi0ForCausalLM
    d = os.path.join(tempfile.mkdtemp(), "tiny_siglip")
    SiglipImageProcessor(size={"height": cfg.v_image, "width": cfg.v_image}).save_pretrained(d)
    vc = dict(model_type="siglip_vision_model", hidden_size=cfg.v_hidden, intermediate_size=cfg.v_inter,
              num_hidden_layers=cfg.v_layers, num_attention_heads=cfg.v_heads, image_size=cfg.v_image,
              patch_size=cfg.v_patch, layer_norm_eps=cfg.v_eps)
    g = dict(model_type="gemma", vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size,
             intermediate_size=cfg.intermediate_size, num_hidden_layers=cfg.num_hidden_layers,
             num_attention_heads=cfg.num_attention_heads, num_key_value_heads=cfg.num_key_value_heads,
             head_dim=cfg.head_dim, max_position_embeddings=512, rope_theta=cfg.rope_theta, rms_norm_eps=cfg.rms_norm_eps)
    a = dict(g)
    a.update(hidden_size=cfg.a_hidden, intermediate_size=cfg.a_inter)
    c = Pi0Config(vision_config=vc, processor_config=d, action_config=a, llm_config=g, mm_projector_type="linear",
                  action_dim=cfg.action_dim, chunk_size=cfg.chunk_size)
    m = Pi0ForCausalLM(c)
    from oracle.pi0_oracle import pi0_shapes
    ref_shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    mine = pi0_shapes(cfg)
    assert ref_shapes == mine, (set(ref_shapes) ^ set(mine))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()}, strict=True)
    for mod in (m.model.llm, m.model.action_expert):           # shim (i): single sqrt(d) scale
        if hasattr(mod.embed_tokens, "embed_scale"):
            mod.embed_tokens.embed_scale.fill_(1.0)
    if not hasattr(DynamicCache, "key_cache"):                 # shim (ii)
        DynamicCache.key_cache = property(lambda self: [l.keys for l in self.layers])
        DynamicCache.value_cache = property(lambda self: [l.values for l in self.layers])
    for p_ in m.parameters():
        p_.requires_grad = True
    return m


class inject:
    """replace torch.normal / Beta.sample inside the reference's forward / inference_action"""

    def __init__(self, noise, time=None):
        self.noise, self.time = noise, time

    def __enter__(self):
        self._normal, self._beta = torch.normal, torch.distributions.Beta.sample
        noise, time = self.noise, self.time
        torch.normal = lambda *a, **k: noise.clone()
        if time is not None:
            # forward computes Beta.sample(shape) * 0.999 + 0.001 (pi0_arch.py:345-351): inject the pre-affine draw
            torch.distributions.Beta.sample = lambda self_, shape=torch.Size(): (time.clone() - 0.001) / 0.999
        return self

    def __exit__(self, *exc):
        torch.normal = self._normal
        torch.distributions.Beta.sample = self._beta


def main():
    sys.path.insert(0, REF)
    sys.path.insert(0, ROOT)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    from oracle.gen_golden import install_timm_shim
    install_timm_shim()
    from oracle.pi0_oracle import Pi0OracleConfig, pi0_shapes
    from oracle.weights import make_weights, weights_crc
    cfg = Pi0OracleConfig()
    seed = 2468
    w = make_weights(pi0_shapes(cfg), seed)
    m = build_reference(cfg, w)
    m.train()
    rs = np.random.RandomState(seed + 1)
    B, L, CAM = 3, 7, 3
    ids = rs.randint(5, cfg.vocab_size - 5, size=(B, L)).astype(np.int64)
    mask = np.ones((B, L), dtype=bool)
    mask[1, 5:] = False                                       # right-padded instruction
    image_masks = np.ones((B, CAM), dtype=bool)
    image_masks[2, 1] = False                                 # a missing camera
    images = np.clip(rs.standard_normal((B, CAM, 3, cfg.v_image, cfg.v_image)), -2.5, 2.5).astype(np.float32)
    states = rs.standard_normal((B, cfg.action_dim)).astype(np.float32)
    actions = rs.standard_normal((B, cfg.chunk_size, cfg.action_dim)).astype(np.float32)
    noise = rs.standard_normal((B, cfg.chunk_size, cfg.action_dim)).astype(np.float32)
    time = rs.uniform(0.05, 0.95, size=(B,)).astype(np.float32)
    t = torch.from_numpy
    with inject(t(noise), t(time)):
        out = m(input_ids=t(ids), attention_mask=t(mask), images=t(images), image_masks=t(image_masks),
                states=t(states), actions=t(actions))
    out.loss.backward()
    sd = dict(m.named_parameters())
    res = dict(weights_crc=np.uint32(weights_crc(w)), seed=np.int64(seed), input_ids=ids, attention_mask=mask,
               images=images, image_masks=image_masks, states=states, actions=actions, noise=noise, time=time,
               loss=np.float32(out.loss.item()), v_t=out.logits.detach().numpy().astype(np.float32))
    res_nograd = [n for n, p_ in sd.items() if p_.grad is None]
    print("[gen_golden_pi0] parameters without a gradient:", res_nograd)
    gsq = 0.0
    for n, p_ in sd.items():
        if p_.grad is not None:
            gsq += float(p_.grad.double().pow(2).sum())
            res["gradN/" + n] = np.float64(p_.grad.double().norm().item())
    res["grad_norm"] = np.float64(gsq ** 0.5)
    for n in ("model.action_out_proj.weight", "model.action_expert.layers.0.self_attn.q_proj.weight",
              "model.llm.layers.0.mlp.down_proj.weight", "model.llm.layers.1.self_attn.k_proj.weight", "model.mm_projector.weight", "model.state_proj.bias",
              "model.llm.layers.0.input_layernorm.weight"):
        res["grad/" + n] = sd[n].grad.numpy().astype(np.float32)
    # inference: injected initial noise, 10 Euler steps
    m.eval()
    init = rs.standard_normal((B, cfg.chunk_size, cfg.action_dim)).astype(np.float32)
    with torch.no_grad(), inject(t(init)):
        acts = m.inference_action(input_ids=t(ids), attention_mask=t(mask), states=t(states), images=t(images),
                                  image_masks=t(image_masks), diffusion_steps=10)
    res["init_noise"] = init
    res["infer_actions"] = acts.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(GOLD, "pi0_t1.npz"), **res)
    print(f"[gen_golden_pi0] loss {res['loss']:.6f} |g| {res['grad_norm']:.4f} infer |a| {np.abs(res['infer_actions']).mean():.4f}")


if __name__ == "__main__":
    main()
