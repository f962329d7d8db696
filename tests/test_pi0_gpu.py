"""Row A11 on the GPU: the native pi0 policy (SigLIP tower, dual-expert Gemma mixture with block-prefix masked
attention, flow-matching head, KV-cached Euler sampler) against golden vectors from the reference Pi0ForCausalLM
(tests/golden/pi0_t1.npz, made by oracle/gen_golden_pi0.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import pi0_oracle as P
from oracle.weights import make_weights, weights_crc

from .helpers import assert_chunk_close, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"
FP32_TOL = 1e-3      # north-star tolerance (observed ~1e-5)


def T(a):
    return torch.from_numpy(np.asarray(a)).to(DEV)


def build(golden_dir, dtype, train=False, fixture="pi0_t1.npz", c=None):
    from dexbotic_amd.model.pi0.pi0_arch import Pi0Config, Pi0ForCausalLM
    g = np.load(os.path.join(golden_dir, fixture), allow_pickle=False)
    c = c or P.Pi0OracleConfig()
    w = make_weights(P.pi0_shapes(c), int(g["seed"]))
    assert weights_crc(w) == int(g["weights_crc"])
    gem = dict(model_type="gemma", vocab_size=c.vocab_size, hidden_size=c.hidden_size, intermediate_size=c.intermediate_size,
               num_hidden_layers=c.num_hidden_layers, num_attention_heads=c.num_attention_heads,
               num_key_value_heads=c.num_key_value_heads, head_dim=c.head_dim, rope_theta=c.rope_theta,
               rms_norm_eps=c.rms_norm_eps)
    act = dict(gem, hidden_size=c.a_hidden, intermediate_size=c.a_inter)
    vis = dict(model_type="siglip_vision_model", hidden_size=c.v_hidden, intermediate_size=c.v_inter,
               num_hidden_layers=c.v_layers, num_attention_heads=c.v_heads, image_size=c.v_image, patch_size=c.v_patch,
               layer_norm_eps=c.v_eps)
    cfg = Pi0Config(vision_config=vis, action_config=act, llm_config=gem, mm_projector_type="linear",
                    action_dim=c.action_dim, chunk_size=c.chunk_size, compute_dtype=dtype)
    m = Pi0ForCausalLM(cfg, device=DEV, train=train)
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == {k: tuple(v.shape) for k, v in w.items()}
    m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=True)
    m.eval()
    return g, m


def test_fp32_pi0_inference_action_matches_reference(golden_dir):
    g, m = build(golden_dir, "float32")
    acts = m.inference_action(input_ids=T(g["input_ids"]), attention_mask=T(g["attention_mask"]), states=T(g["states"]),
                              images=T(g["images"]), image_masks=T(g["image_masks"]), diffusion_steps=10,
                              noise=T(g["init_noise"]))
    assert tuple(acts.shape) == g["infer_actions"].shape
    assert rel_err(acts.cpu().numpy(), g["infer_actions"]) < FP32_TOL
    assert_chunk_close(acts.cpu().numpy(), g["infer_actions"], what="pi0 chunk")


def test_pi0_sampler_graph_replay_equals_eager_launches(golden_dir):
    """the Euler loop is captured into one HIP graph per shape (graphs.GraphCache: 1st call eager, 2nd captures, then replays);
    every call gets its own initial noise and must equal the eager (use_graph=False) result bit for bit — fp32 and bf16"""
    for dtype in ("float32", "bfloat16"):
        g, m = build(golden_dir, dtype)
        kw = dict(input_ids=T(g["input_ids"]), attention_mask=T(g["attention_mask"]), states=T(g["states"]),
                  images=T(g["images"]), image_masks=T(g["image_masks"]), diffusion_steps=10)
        for i in range(4):
            noise = torch.from_numpy(np.random.RandomState(50 + i).standard_normal(g["init_noise"].shape).astype(np.float32)).to(DEV)
            want = m.inference_action(noise=noise, use_graph=False, **kw)
            got = m.inference_action(noise=noise, use_graph=True, **kw)
            assert torch.equal(got, want), (dtype, i)
        assert any(e["graph"] is not None for e in m._sampler_graphs.entries.values())
        if dtype == "float32":
            acts = m.inference_action(noise=T(g["init_noise"]), use_graph=True, **kw)
            assert rel_err(acts.cpu().numpy(), g["infer_actions"]) < FP32_TOL


def test_fp32_pi0_forward_loss_matches_reference(golden_dir):
    g, m = build(golden_dir, "float32")
    with torch.no_grad():
        out = m(input_ids=T(g["input_ids"]), attention_mask=T(g["attention_mask"]), images=T(g["images"]),
                image_masks=T(g["image_masks"]), states=T(g["states"]), actions=T(g["actions"]), noise=T(g["noise"]),
                time=g["time"])
    assert rel_err(out.logits.cpu().numpy(), g["v_t"]) < FP32_TOL
    assert abs(out.loss.item() - float(g["loss"])) < FP32_TOL * abs(float(g["loss"]))


def test_fp32_pi0_training_step_grads_match_reference(golden_dir):
    g, m = build(golden_dir, "float32", train=True)
    m.train()
    st = m.store
    st.set_expected(m.unused_parameter_names())
    st.begin_step()
    out = m(input_ids=T(g["input_ids"]), attention_mask=T(g["attention_mask"]), images=T(g["images"]),
            image_masks=T(g["image_masks"]), states=T(g["states"]), actions=T(g["actions"]), noise=T(g["noise"]),
            time=g["time"])
    assert abs(out.loss.item() - float(g["loss"])) < FP32_TOL * abs(float(g["loss"]))
    out.loss.backward()
    for key in g.files:
        if key.startswith("grad/"):
            assert rel_err(st.g(key[5:]).cpu().numpy(), g[key]) < FP32_TOL, key
        elif key.startswith("gradN/"):
            gn = float(g[key])
            assert st.grad_written[key[6:]], key
            assert abs(st.g(key[6:]).double().norm().item() - gn) < FP32_TOL * gn + 1e-6 * float(g["grad_norm"]), key
    # parameters the reference leaves without a gradient are exactly the ones reported unused
    have = {k[6:] for k in g.files if k.startswith("gradN/")}
    assert set(m.unused_parameter_names()) == set(st.slots) - have


def test_bf16_pi0_inference_tracks_reference(golden_dir):
    g, m = build(golden_dir, "bfloat16")
    acts = m.inference_action(input_ids=T(g["input_ids"]), attention_mask=T(g["attention_mask"]), states=T(g["states"]),
                              images=T(g["images"]), image_masks=T(g["image_masks"]), diffusion_steps=10,
                              noise=T(g["init_noise"]))
    # the reference samples in fp32 (pi0_exp.py:347-353); bf16 compute is the training dtype.  Ten Euler steps through a
    # random-weight tiny model amplify rounding, so this is a sanity bound on the relative L2 error, not a parity claim (the bf16
    # parity claim is the real-width test below, held to the reference under autocast).  Measured 5.5e-2 on the MI355X: bound 1.5 x that
    a, r = acts.cpu().numpy().astype(np.float64), g["infer_actions"].astype(np.float64)
    d = np.linalg.norm(a - r) / np.linalg.norm(r)
    print(f"toy pi0 bf16 sampler vs fp32 reference: relative L2 {d:.3e}")
    assert d < 8.5e-2


def test_bf16_siglip_tower_head_dim_72_padded_attention():
    """SigLIP-So400m has head_dim 72: the bf16 tower runs the MFMA attention kernels on their 128-wide tiles with the 72 real
    columns only (round 5: no zero-padded copies of q / k / v / o any more — attention.hip, DV).
    Forward features and the gradient of a scalar against the fp32 CPU restatement."""
    from dexbotic_amd.engine import ParamStore, attach_parameters
    from dexbotic_amd.model.modules.mm_vision.siglip.siglip_encoder import SiglipVisionConfig, SiglipVisionTower
    c = P.Pi0OracleConfig(v_hidden=144, v_inter=192, v_layers=2, v_heads=2, v_image=56, v_patch=14)
    shapes = {k: v for k, v in P.pi0_shapes(c).items() if k.startswith("model.mm_vision_tower.")}
    w = make_weights(shapes, 99)
    st = ParamStore(DEV, torch.bfloat16)
    tower = SiglipVisionTower(SiglipVisionConfig(hidden_size=144, intermediate_size=192, num_hidden_layers=2,
                                                 num_attention_heads=2, image_size=56, patch_size=14), st)
    st.finalize(train=True)
    attach_parameters(tower, st)
    for k, v in w.items():
        st.w32(k).copy_(torch.from_numpy(v).to(DEV))
    st.sync_shadow()
    st.set_expected(tower.unused_parameter_names())
    st.begin_step()
    imgs = torch.from_numpy(np.clip(np.random.RandomState(3).standard_normal((3, 3, 56, 56)), -2.5, 2.5).astype(np.float32))
    out = tower(imgs.to(DEV))
    sd = {k: torch.from_numpy(v).requires_grad_(True) for k, v in w.items()}
    ref = P.siglip_features(sd, c, imgs)
    assert rel_err(out.float().detach().cpu().numpy(), ref.detach().numpy()) < 4e-2
    probe = torch.from_numpy(np.random.RandomState(4).standard_normal(tuple(ref.shape)).astype(np.float32))
    (ref * probe).sum().backward()
    (out.float() * probe.to(DEV)).sum().backward()
    for n in ("model.mm_vision_tower.vision_tower.encoder.layers.0.self_attn.q_proj.weight",
              "model.mm_vision_tower.vision_tower.encoder.layers.1.self_attn.v_proj.weight",
              "model.mm_vision_tower.vision_tower.embeddings.position_embedding.weight"):
        gr = sd[n].grad.double()
        assert (st.g(n).double().cpu() - gr).norm() / gr.norm() < 5e-2, n


# ------------------------------------------------------------------------------------------------------------------
# REAL widths, pinned to the reference's own Pi0ForCausalLM (tests/golden/pi0_real_ref.npz, oracle/gen_golden_pi0_real.py):
# Gemma-2B expert (d 2048, ffn 16384, 8 q / 1 kv heads x 256) + action expert (d 1024), SigLIP-So400m tower (d 1152, 16 heads
# x 72 -> the hd-128 kernels on zero-padded heads), 2 mixture + 2 tower layers, chunk 50 (the reference default), prefix 784.
# The hd-256 block-masked flash kernels (forward + fused backward) run at the shapes bench.py's secondary pi0 line times.
from oracle import gen_golden_pi0_real as PR


def _real_inputs(g):
    import zlib
    x = PR.inputs()
    assert zlib.crc32(x["images"].tobytes()) == int(g["images_crc"])
    return x


def _pi0_step(m, x):
    st = m.store
    st.set_expected(m.unused_parameter_names())
    st.begin_step()
    out = m(input_ids=T(x["input_ids"]), attention_mask=T(x["attention_mask"]), images=T(x["images"]),
            image_masks=T(x["image_masks"]), states=T(x["states"]), actions=T(x["actions"]), noise=T(x["noise"]),
            time=x["time"])
    out.loss.backward()
    torch.cuda.synchronize()
    res = {"loss": out.loss.item(), "v_t": out.logits.detach().float().cpu().numpy()}
    for name, pre in PR.GROUPS.items():
        sq = sum(float(st.g(n).double().pow(2).sum()) for n in st.slots if n.startswith(pre) and st.grad_written[n])
        res[f"gnorm/{name}"] = sq ** 0.5
    for n in PR.GSAMP:
        res["gsamp/" + n] = st.g(n).reshape(-1)[::PR.STRIDE].float().cpu().numpy()
    return res


def test_fp32_pi0_real_width_step_and_chunk50_inference_match_reference_classes(golden_dir):
    g, m = build(golden_dir, "float32", train=True, fixture="pi0_real_ref.npz", c=PR.REAL)
    x = _real_inputs(g)
    m.train()
    got = _pi0_step(m, x)
    for k, v in got.items():
        d = rel_err(v, g["fp32/" + k])
        assert d < FP32_TOL, (k, d)
    m.eval()
    # the reference samples in fp32 (pi0_exp.py:347-353) with its default chunk_size 50 (pi0_arch.py:58-59)
    acts = m.inference_action(input_ids=T(x["input_ids"]), attention_mask=T(x["attention_mask"]), states=T(x["states"]),
                              images=T(x["images"]), image_masks=T(x["image_masks"]), diffusion_steps=10,
                              noise=T(x["init_noise"]))
    assert tuple(acts.shape) == g["fp32/infer_actions"].shape == (2, 50, 32)
    assert rel_err(acts.cpu().numpy(), g["fp32/infer_actions"]) < FP32_TOL
    # (atol: ten Euler steps through 18 real-width layers leave 2.0e-5 on a component of magnitude 3e-3 — 8e-6 of the chunk's largest
    #  component, 2.6; every other fixture passes the helper's default 1e-5)
    assert_chunk_close(acts.cpu().numpy(), g["fp32/infer_actions"], atol=5e-5, what="pi0 real-width chunk 50")


# bf16 compute vs the reference under torch.autocast("cpu", bfloat16) (HF Trainer bf16=True).  The yardstick for "how far apart
# may two bf16 evaluations of this stack be" comes from the reference itself: its autocast run sits 4.6e-2 (v_t) and up to 1.1e-1
# (strided gradient samples, max-norm relative) from its own fp32 run (Gemma's sqrt(d) = 45x embedding scale and head_dim 256
# make this stack loud in bf16).  Element-wise quantities: within 2x THAT distance of the bf16 reference (observed on the
# MI355X: v_t 5.7e-2, gradient samples <= 1.14e-1, i.e. ~1.0-1.2x).  Scalars: loss 3e-3 (observed 9e-5), per-group gradient norms
# 1e-2 (observed <= 6.4e-3; the reference's own bf16-vs-fp32 norms differ by <= 1.6e-3 — the product keeps bf16 activations for
# the backward where autocast keeps fp32 residuals).
PI0_BF16_SCALAR = {"loss": 3e-3, "gnorm": 1e-2}


def test_bf16_pi0_real_width_step_tracks_reference_under_autocast(golden_dir):
    g, m = build(golden_dir, "bfloat16", train=True, fixture="pi0_real_ref.npz", c=PR.REAL)
    x = _real_inputs(g)
    m.train()
    got = _pi0_step(m, x)
    worst = {}
    print("bf16 pi0 product vs reference-under-autocast | vs fp32 reference:")
    for k, v in got.items():
        d = rel_err(v, g["bf16/" + k])
        print(f"  {k:75s} {d:.2e} | {rel_err(v, g['fp32/' + k]):.2e}")
        scal = [b for pre, b in PI0_BF16_SCALAR.items() if k.startswith(pre)]
        bound = scal[0] if scal else 2.0 * rel_err(g["bf16/" + k], g["fp32/" + k])
        if d >= bound:
            worst[k] = (d, bound)
    assert not worst, worst
