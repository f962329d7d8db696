"""Row A10 at the BASELINE widths: lm_head + cross-entropy training step and KV-cached greedy decode of the native
DexboticForCausalLM against the REFERENCE class at d 3584 / ffn 18944 / 28 q 4 kv heads x 128 / CLIP-L, four decoder layers
(tests/golden/lm_real_ref.npz, oracle/gen_golden_lm_real.py)."""
import os

import numpy as np
import pytest
import torch

from oracle.gen_golden_lm_real import GROUPS, GSAMP, REAL4_LM, STRIDE, lm_weights
from oracle.weights import weights_crc

from .helpers import build_lm_product, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"
FP32_TOL = 1e-3


def T(a):
    return torch.from_numpy(np.asarray(a)).to(DEV)


@pytest.fixture(scope="module")
def fixture(golden_dir):
    g = np.load(os.path.join(golden_dir, "lm_real_ref.npz"), allow_pickle=False)
    w = lm_weights()
    assert weights_crc(w) == int(g["weights_crc"])
    return g, w


def _step(m, g):
    m.train()
    m.store.begin_step()
    out = m(input_ids=T(g["input_ids"]), attention_mask=T(g["attention_mask"]), labels=T(g["labels"]), images=T(g["images"]))
    out.loss.backward()
    st = m.store
    res = {"loss": out.loss.item(), "logits": out.logits.detach().float().cpu().numpy()[:, g["positions"]]}
    gsq = {k: 0.0 for k in GROUPS}
    for n in st.slots:
        if not st.grad_written.get(n, False):
            continue
        for k, pre in GROUPS.items():
            if n.startswith(pre):
                gsq[k] += float(st.g(n).double().pow(2).sum())
    for k in GROUPS:
        res["gnorm/" + k] = gsq[k] ** 0.5
    for n in GSAMP:
        res["gsamp/" + n] = st.g(n).float().reshape(-1)[::STRIDE].cpu().numpy()
    return res


def test_fp32_lm_step_at_real_widths_matches_the_reference_class(fixture):
    g, w = fixture
    m = build_lm_product(REAL4_LM, w, "float32", DEV, train=True)
    res = _step(m, g)
    assert abs(res["loss"] - float(g["fp32/loss"])) < FP32_TOL * abs(float(g["fp32/loss"]))
    assert rel_err(res["logits"], g["fp32/logits"]) < FP32_TOL
    for k in GROUPS:
        assert abs(res["gnorm/" + k] - float(g["fp32/gnorm/" + k])) < FP32_TOL * float(g["fp32/gnorm/" + k]), k
    for n in GSAMP:
        assert rel_err(res["gsamp/" + n], g["fp32/gsamp/" + n]) < FP32_TOL, n


def test_bf16_lm_step_at_real_widths_tracks_the_reference_under_autocast(fixture):
    """bf16 compute against the reference under autocast(bfloat16): within 2 x the reference's own bf16-vs-fp32 distance of its bf16
    run (and of its fp32 run 3 x)"""
    g, w = fixture
    m = build_lm_product(REAL4_LM, w, "bfloat16", DEV, train=True)
    res = _step(m, g)
    bad = {}
    for k in ["loss", "logits"] + ["gnorm/" + x for x in GROUPS]:
        gap = max(float(g["ref_bf16_vs_fp32/" + k]), 2e-4)
        d16, d32 = rel_err(res[k], g["bf16/" + k]), rel_err(res[k], g["fp32/" + k])
        print(f"  {k:18s} vs reference bf16 {d16:.2e} | vs reference fp32 {d32:.2e} | reference bf16 vs fp32 {gap:.2e}")
        if d16 > 2.0 * gap or d32 > 3.0 * gap:
            bad[k] = (d16, d32, gap)
    assert not bad, bad


@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
def test_greedy_decode_at_real_widths_reproduces_the_reference_ids(fixture, dtype):
    """KV-cached decode (fp32: the per-op step; bf16: the persistent one-launch step) against the reference's full-prefix greedy loop:
    ids exact wherever the reference's top-1 / top-2 margin exceeds twice the distance between the two logit rows"""
    g, w = fixture
    m = build_lm_product(REAL4_LM, w, dtype, DEV, train=False)
    m.eval()
    n_new = len(g["decode_new_ids"])
    out = m.generate(T(g["decode_prompt"]), images=T(g["images"][:1]), max_new_tokens=n_new, do_sample=False,
                     return_dict_in_generate=True, output_logits=True)
    L0 = g["decode_prompt"].shape[1]
    got = out.sequences[0, L0:].cpu().numpy()
    tol = FP32_TOL if dtype == "float32" else 3.0 * float(g["ref_bf16_vs_fp32/logits"])
    for t in range(n_new):
        lg = out.logits[t][0].float().cpu().numpy()
        dist = float(np.abs(lg - g["decode_logits"][t]).max())
        assert rel_err(lg, g["decode_logits"][t]) < tol, (t, rel_err(lg, g["decode_logits"][t]))
        margin = float(g["decode_margin"][t])
        print(f"  {dtype} token {t}: id {got[t]} / {g['decode_new_ids'][t]}  margin {margin:.4f}  logit distance {dist:.2e}")
        if margin > 2.0 * dist:
            assert got[t] == g["decode_new_ids"][t], t
        if got[t] != g["decode_new_ids"][t]:
            break                                            # a near-tie decided the other way: later rows see another prefix
    if dtype == "bfloat16":
        from dexbotic_amd import kernels as K
        assert not K.decode_timed_out()
