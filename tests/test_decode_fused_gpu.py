"""Row A10, decode step: the persistent one-launch decoder pass of a new token (csrc/decode_fused.hip, Qwen2Backbone.forward_cached
with batch 1 / one token / bf16) against the per-op decode step it replaces and against the same model computed in fp32.
The golden greedy ids of the reference are held by tests/test_lm_gpu.py (fp32) — here the two bf16 paths are held to each other:
same rounding points, different summation order, fp32 attention instead of bf16 probabilities."""
import numpy as np
import pytest
import torch

from oracle import cogact_oracle as O
from oracle.weights import cogact_shapes, make_weights

from .helpers import CFGS, build_lm_product, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _lm_weights(cfg, seed=77):
    return {k: v for k, v in make_weights(cogact_shapes(cfg), seed).items() if ".action_head." not in k}


def _bf16_round(w):
    return {k: torch.from_numpy(v).to(torch.bfloat16).float().numpy() for k, v in w.items()}


def _walk(m, prompt, steps, pad, dtype):
    """prefill + one pass per new embedding; returns (hidden state per step [T, d], cache)"""
    llm = m.model.llm
    cache = llm.new_cache(1, prompt.shape[1] + len(steps), DEV, dtype)
    llm.forward_cached(prompt.to(dtype), cache, pad)
    out = [llm.forward_cached(e.to(dtype).view(1, 1, -1), cache, pad)[0, 0].float().cpu().numpy() for e in steps]
    torch.cuda.synchronize()
    return np.stack(out), cache


@pytest.mark.parametrize("tag,pad", [("t1", None), ("t2", None), ("t2", [3])])
def test_persistent_decode_step_tracks_the_per_op_step_and_the_fp32_model(tag, pad, monkeypatch):
    cfg = CFGS[tag]
    w = _lm_weights(cfg)
    m16 = build_lm_product(cfg, w, "bfloat16", DEV, train=False)
    m32 = build_lm_product(cfg, _bf16_round(w), "float32", DEV, train=False)      # the same weight VALUES, fp32 arithmetic
    g = torch.Generator().manual_seed(5)
    d = cfg.hidden_size
    prompt = (torch.randn(1, 11, d, generator=g) * 0.5).to(DEV)
    steps = [(torch.randn(d, generator=g) * 0.5).to(DEV) for _ in range(6)]
    prompt, steps = prompt.bfloat16().float(), [e.bfloat16().float() for e in steps]     # every path sees the same bf16 inputs
    ref, _ = _walk(m32, prompt, steps, pad, torch.float32)
    monkeypatch.setenv("DXA_DECODE_FUSED", "0")
    per_op, c_u = _walk(m16, prompt, steps, pad, torch.bfloat16)
    monkeypatch.setenv("DXA_DECODE_FUSED", "1")
    fused, c_f = _walk(m16, prompt, steps, pad, torch.bfloat16)
    assert c_u.fused_steps == 0 and c_f.fused_steps == len(steps)
    from dexbotic_amd import kernels as K
    assert not K.decode_timed_out()
    for t in range(len(steps)):
        e_f, e_u, e_fu = rel_err(fused[t], ref[t]), rel_err(per_op[t], ref[t]), rel_err(fused[t], per_op[t])
        print(f"{tag} pad {pad} step {t}: fused vs fp32 {e_f:.2e} | per-op vs fp32 {e_u:.2e} | fused vs per-op {e_fu:.2e}")
        assert e_f < 3e-2 and e_f <= 1.5 * e_u + 3e-3, (t, e_f, e_u)
    n0 = prompt.shape[1]
    for i in range(cfg.num_hidden_layers):                  # the appended keys / values: what later tokens attend to
        for a, b in ((c_f.k[i], c_u.k[i]), (c_f.v[i], c_u.v[i])):
            assert rel_err(a[:, :, n0:c_f.length].float().cpu().numpy(), b[:, :, n0:c_u.length].float().cpu().numpy()) < 3e-2, i


def test_persistent_decode_step_at_the_7b_widths(monkeypatch):
    """two decoder layers at the real widths (3584 / 18944, 28 query heads over 4 key heads, head_dim 128: 18 / 14 / 74 / 14 rows per
    workgroup on 256 CUs, 7 and 37 K blocks of 512) over a 300-token cache"""
    cfg = O.OracleConfig(vocab_size=512, hidden_size=3584, intermediate_size=18944, num_hidden_layers=2, num_attention_heads=28,
                         num_key_value_heads=4)
    m16 = build_lm_product(cfg, _lm_weights(cfg, seed=3), "bfloat16", DEV, train=False)
    g = torch.Generator().manual_seed(9)
    prompt = (torch.randn(1, 300, cfg.hidden_size, generator=g) * 0.5).to(DEV)
    steps = [(torch.randn(cfg.hidden_size, generator=g) * 0.5).to(DEV) for _ in range(3)]
    monkeypatch.setenv("DXA_DECODE_FUSED", "0")
    per_op, c_u = _walk(m16, prompt, steps, None, torch.bfloat16)
    monkeypatch.setenv("DXA_DECODE_FUSED", "1")
    fused, c_f = _walk(m16, prompt, steps, None, torch.bfloat16)
    assert c_u.fused_steps == 0 and c_f.fused_steps == 3
    for t in range(3):
        e = rel_err(fused[t], per_op[t])
        print(f"7B widths step {t}: fused vs per-op {e:.2e}")
        assert e < 2e-2, (t, e)
    again, _ = _walk(m16, prompt, steps, None, torch.bfloat16)
    assert np.array_equal(again, fused)                      # fixed summation order: run-to-run identical


def test_bf16_generate_uses_the_persistent_step_and_agrees_with_the_per_op_loop(golden_dir, monkeypatch):
    """generate() in bf16: greedy ids of the persistent path = ids of the per-op path wherever the per-op path's top-1 / top-2
    logit margin exceeds the distance between the two paths' logits"""
    from .helpers import load_lm_golden
    g, cfg, w = load_lm_golden(golden_dir)
    m = build_lm_product(cfg, w, "bfloat16", DEV, train=False)
    m.eval()
    ids = torch.from_numpy(g["decode_prompt"]).to(DEV)
    img = torch.from_numpy(g["images"][:1]).to(DEV)
    outs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("DXA_DECODE_FUSED", mode)
        outs[mode] = m.generate(ids, images=img, max_new_tokens=6, do_sample=False, return_dict_in_generate=True, output_logits=True)
    L0 = ids.shape[1]
    a, b = outs["0"].sequences[0, L0:].cpu().numpy(), outs["1"].sequences[0, L0:].cpu().numpy()
    for t in range(len(a)):
        la, lb = outs["0"].logits[t][0].float().cpu().numpy(), outs["1"].logits[t][0].float().cpu().numpy()
        top2 = np.sort(la)[-2:]
        margin, dist = float(top2[1] - top2[0]), float(np.abs(la - lb).max())
        print(f"token {t}: ids {a[t]} / {b[t]}  margin {margin:.3f}  logit distance {dist:.3f}")
        if margin > 2 * dist:
            assert a[t] == b[t], t
        if a[t] != b[t]:
            break                                           # the sequences part ways at a near-tie: later steps see different prefixes
