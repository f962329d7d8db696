"""Row A10 on the GPU: lm_head + causal-LM cross-entropy training step and KV-cached greedy decode of the native
DexboticForCausalLM / DiscreteVLAForCausalLM against golden vectors from the reference (tests/golden/lm_t1.npz)."""
import numpy as np
import pytest
import torch

from .helpers import build_lm_product, load_lm_golden, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"
FP32_TOL = 1e-3      # north-star tolerance (observed ~1e-5)


def T(a):
    return torch.from_numpy(np.asarray(a)).to(DEV)


def test_fp32_lm_loss_and_grads_match_reference(golden_dir):
    g, cfg, w = load_lm_golden(golden_dir)
    m = build_lm_product(cfg, w, "float32", DEV, train=True)
    m.train()
    m.store.begin_step()
    out = m(input_ids=T(g["input_ids"]), attention_mask=T(g["attention_mask"]), labels=T(g["labels"]), images=T(g["images"]))
    assert rel_err(out.logits.detach().cpu().numpy(), g["logits"]) < FP32_TOL
    assert abs(out.loss.item() - float(g["loss"])) < FP32_TOL * abs(float(g["loss"]))
    out.loss.backward()
    st = m.store
    for key in g.files:
        if key.startswith("grad/"):
            assert rel_err(st.g(key[5:]).cpu().numpy(), g[key]) < FP32_TOL, key
        elif key.startswith("gradN/") and key[6:] in st.slots and st.grad_written.get(key[6:], False):
            gn = float(g[key])
            # (k_proj biases have a mathematically zero gradient — softmax shift invariance — hence the floor)
            assert abs(st.g(key[6:]).double().norm().item() - gn) < FP32_TOL * gn + 1e-6 * float(g["grad_norm"]), key
    rows = T(g["embed_rows"])
    assert rel_err(st.g("model.llm.embed_tokens.weight")[rows].cpu().numpy(), g["grad_embed_rows"]) < FP32_TOL


@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
def test_lm_head_weight_gradient_written_in_row_slabs(golden_dir, dtype, monkeypatch):
    """the fp32 dW of the full 152064-row vocabulary (2.18 GB) exceeds what one launch of the MFMA fast path addresses: it is
    written in row slabs (column slices of dZ).  With the slab limit shrunk so that the toy vocabulary needs several slabs the
    gradient must equal the single-launch one bit for bit"""
    from dexbotic_amd import functional as Fn
    g, cfg, w = load_lm_golden(golden_dir)
    grads = []
    for slab_bytes in (Fn.LMHEAD_SLAB_BYTES, 256 * 4 * cfg.hidden_size):      # one launch | 256-row slabs
        monkeypatch.setattr(Fn, "LMHEAD_SLAB_BYTES", slab_bytes)
        m = build_lm_product(cfg, w, dtype, DEV, train=True)
        m.train()
        m.store.begin_step()
        out = m(input_ids=T(g["input_ids"]), attention_mask=T(g["attention_mask"]), labels=T(g["labels"]), images=T(g["images"]))
        out.loss.backward()
        torch.cuda.synchronize()
        grads.append(m.store.g("lm_head.weight").clone())
    assert cfg.vocab_size > 256 and torch.equal(grads[0], grads[1])


def test_bf16_lm_loss_tracks_reference(golden_dir):
    g, cfg, w = load_lm_golden(golden_dir)
    m = build_lm_product(cfg, w, "bfloat16", DEV, train=True)
    m.train()
    m.store.begin_step()
    out = m(input_ids=T(g["input_ids"]), attention_mask=T(g["attention_mask"]), labels=T(g["labels"]), images=T(g["images"]))
    assert abs(out.loss.item() - float(g["loss"])) < 2e-2 * abs(float(g["loss"]))
    out.loss.backward()
    gn = float(g["gradN/lm_head.weight"])
    assert abs(m.store.g("lm_head.weight").double().norm().item() - gn) < 6e-2 * gn


def test_fp32_greedy_decode_token_ids_exact(golden_dir):
    """KV-cached decode must pick, token for token, what the reference's full-prefix greedy loop picked"""
    g, cfg, w = load_lm_golden(golden_dir)
    m = build_lm_product(cfg, w, "float32", DEV, train=False)
    m.eval()
    n_new = len(g["decode_new_ids"])
    out = m.generate(T(g["decode_prompt"]), images=T(g["images"][:1]), max_new_tokens=n_new, do_sample=False,
                     return_dict_in_generate=True, output_logits=True)
    L0 = g["decode_prompt"].shape[1]
    assert np.array_equal(out.sequences[0, :L0].cpu().numpy(), g["decode_prompt"][0])
    assert np.array_equal(out.sequences[0, L0:].cpu().numpy(), g["decode_new_ids"])          # integer row: exact
    for i, lg in enumerate(out.logits):
        assert rel_err(lg[0].cpu().numpy(), g["decode_logits"][i]) < FP32_TOL, i
    # eos stops the loop early
    eos = int(g["decode_new_ids"][2])
    seq = m.generate(T(g["decode_prompt"]), images=T(g["images"][:1]), max_new_tokens=n_new, eos_token_id=eos)
    assert seq.shape[1] == L0 + 3 and int(seq[0, -1]) == eos


def test_generate_left_padded_unequal_prompts(golden_dir):
    """a batch of two prompts of different length (left padded + attention mask, like HF generate wants them): each sample
    continues exactly as it does alone — per-sample key ranges in the cache and rotary positions counted from the first
    real token; right-padded batches are refused"""
    g, cfg, w = load_lm_golden(golden_dir)
    m = build_lm_product(cfg, w, "float32", DEV, train=False)
    m.eval()
    long_p = g["decode_prompt"][0]
    short_p = np.concatenate([long_p[:-3], long_p[-1:]])          # same image placeholder, three text tokens fewer
    assert (short_p == -200).sum() == (long_p == -200).sum() == 1
    n_new = 6
    img = T(g["images"][:1])
    alone = [m.generate(T(p_[None]), images=img, max_new_tokens=n_new)[0, len(p_):].cpu().numpy() for p_ in (long_p, short_p)]
    pad = len(long_p) - len(short_p)
    ids = np.stack([long_p, np.concatenate([np.zeros(pad, dtype=long_p.dtype), short_p])])
    mask = np.ones_like(ids, dtype=bool)
    mask[1, :pad] = False
    m.config.tokenizer_padding_side = "left"
    seq = m.generate(T(ids), images=torch.cat([img, img]), max_new_tokens=n_new, attention_mask=T(mask))
    assert np.array_equal(seq[0, ids.shape[1]:].cpu().numpy(), alone[0])
    assert np.array_equal(seq[1, ids.shape[1]:].cpu().numpy(), alone[1])
    m.config.tokenizer_padding_side = "right"
    with pytest.raises(ValueError):
        m.generate(T(ids), images=torch.cat([img, img]), max_new_tokens=2, attention_mask=T(mask))


class _FakeTokenizer:
    """ids -> text: token i decodes to ' {i % 255}', id 7 is the stop string '</s>'"""
    bos_token_id = None

    def __call__(self, text):
        class R:
            input_ids = [7]
        return R()

    def decode(self, ids, skip_special_tokens=False):
        return "".join("</s>" if int(i) == 7 else f" {int(i) % 255}" for i in ids)

    def batch_decode(self, ids, skip_special_tokens=True):
        return [self.decode(row) for row in ids]


def test_discrete_vla_inference_action(golden_dir):
    """digit tokens -> bins -> [-1,1] -> de-normalised row: integer/str arithmetic identical to the reference's
    _discrete_action_to_continuous + _denorm on the greedy continuation"""
    from dexbotic_amd.model.discrete_vla.discrete_vla_arch import DiscreteVLAForCausalLM
    from oracle import cogact_oracle as O
    g, cfg, w = load_lm_golden(golden_dir)
    m = build_lm_product(cfg, w, "float32", DEV, train=False, cls=DiscreteVLAForCausalLM)
    m.eval()

    class Conv:
        sep, sep2 = "</s>", "</s>"
        sep_style = type("S", (), {"name": "TWO"})()

    norms = {"min": [-1.0, -2.0, -3.0, -1.0, -1.0, -1.0, 0.0], "max": [1.0, 2.0, 3.0, 1.0, 1.0, 1.0, 1.0]}
    tok = _FakeTokenizer()
    # 6 golden tokens are not enough for 7 bins: let it run on greedily for 8
    acts = m.inference_action(T(g["decode_prompt"]), T(g["images"][:1]),
                              {"conv": Conv(), "tokenizer": tok, "vocab_size": 255, "action_norms": norms,
                               "do_sample": False, "max_new_tokens": 8})
    seq = m.generate(T(g["decode_prompt"]), images=T(g["images"][:1]), max_new_tokens=8)
    new = seq[0, g["decode_prompt"].shape[1]:].cpu().numpy()
    assert np.array_equal(new[:6], g["decode_new_ids"])
    text = tok.decode(new).strip("</s>")
    want = O.denorm(O.discrete_action_to_continuous(text, 255), norms)
    assert np.array_equal(np.asarray(acts), want)


def test_fp32_hybrid_cogact_step_matches_reference(golden_dir):
    """HybridCogACTForCausalLM: text CE + has_action-weighted diffusion loss, one backward (golden hybrid_t1)"""
    import os
    from dexbotic_amd.model.cogact.hybrid_cogact_arch import HybridCogACTForCausalLM
    from oracle.weights import cogact_shapes, make_weights, weights_crc
    from .helpers import CFGS, product_config
    g = np.load(os.path.join(golden_dir, "hybrid_t1.npz"), allow_pickle=False)
    cfg = CFGS["t1"]
    w = make_weights(cogact_shapes(cfg), int(g["seed"]))
    assert weights_crc(w) == int(g["weights_crc"])
    m = HybridCogACTForCausalLM(product_config(cfg, "float32"), device=DEV, train=True)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=True)
    m.train()
    st = m.store
    st.set_expected(m.unused_parameter_names())
    st.begin_step()
    out = m(input_ids=T(g["input_ids"]), attention_mask=T(g["attention_mask"]), labels=T(g["labels"]), images=T(g["images"]),
            actions=T(g["actions"]), has_action=T(g["has_action"]), has_text=T(g["has_text"]), noise=T(g["noise"]),
            timesteps=T(g["timesteps"]), drop_ids=T(g["drop_u"]) < 0.1)
    for k in ("loss", "text_loss", "action_loss"):
        assert abs(getattr(out, k).item() - float(g[k])) < FP32_TOL * abs(float(g[k])), k
    out.loss.backward()
    for key in g.files:
        if key.startswith("grad/"):
            assert rel_err(st.g(key[5:]).cpu().numpy(), g[key]) < FP32_TOL, key
        elif key.startswith("gradN/") and st.grad_written.get(key[6:], False):
            gn = float(g[key])
            assert abs(st.g(key[6:]).double().norm().item() - gn) < FP32_TOL * gn + 1e-6 * float(g["grad_norm"]), key
