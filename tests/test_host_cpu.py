"""CPU tests of the host side of the product: construction, state_dict key map, splice plan, diffusion
tables, integer action rows, optimizer grouping, registry.  No kernel is launched here."""
import os

import numpy as np
import pytest
import torch

from oracle import cogact_oracle as O
from oracle.weights import cogact_shapes
from tests.helpers import CFGS, build_product, load_golden, product_config


@pytest.mark.parametrize("tag", ["t1", "t2"])
def test_state_dict_key_map_matches_reference(golden_dir, tag):
    # oracle.weights.cogact_shapes is asserted equal to the live reference's state_dict by gen_golden.py
    g, cfg, w = load_golden(golden_dir, tag)
    m = build_product(cfg, w, device="cpu", train=False)
    sd = m.state_dict()
    assert {k: tuple(v.shape) for k, v in sd.items()} == cogact_shapes(cfg)
    for k, v in w.items():
        assert np.array_equal(sd[k].numpy(), v), k
    # every parameter is a view into ONE arena; fused groups are adjacent
    st = m.store
    base = st.master.data_ptr()
    for name, p in m.named_parameters():
        assert base <= p.data_ptr() < base + st.master.numel() * 4, name
    lp = "model.llm.layers.0.self_attn."
    fused = st.w32(lp + "q_proj.weight", lp + "k_proj.weight", lp + "v_proj.weight",
                   shape=((cfg.num_attention_heads + 2 * cfg.num_key_value_heads) * cfg.head_dim, cfg.hidden_size))
    assert torch.equal(fused[:cfg.num_attention_heads * cfg.head_dim], sd[lp + "q_proj.weight"])
    assert torch.equal(fused[-cfg.num_key_value_heads * cfg.head_dim:], sd[lp + "v_proj.weight"])
    # the 4.51-style checkpoint key layout is accepted too
    old = {k.replace(".vision_tower.", ".vision_tower.vision_model."): torch.from_numpy(v) for k, v in w.items()}
    m.load_state_dict(old, strict=True)


def test_unused_parameters_match_reference(golden_dir):
    g, cfg, w = load_golden(golden_dir, "t1")
    m = build_product(cfg, w, device="cpu", train=False)
    assert sorted(m.unused_parameter_names()) == sorted(g["no_grad_params"].tolist())


@pytest.mark.parametrize("tag", ["t1", "t2"])
def test_splice_plan_matches_reference_and_oracle(golden_dir, tag):
    from dexbotic_amd.splice import build_splice_plan
    g, cfg, _ = load_golden(golden_dir, tag)
    views = g["images"].shape[1] if g["images"].ndim == 5 else 1
    n_img = cfg.num_patches * views
    plan = build_splice_plan(g["input_ids"], g["attention_mask"], g["input_ids"], n_img)
    assert np.array_equal(plan.attention_mask, g["new_attention_mask"])          # reference output
    src, mask, lengths = O.splice_plan(g["input_ids"], g["attention_mask"], n_img)
    assert np.array_equal(plan.plan, src) and np.array_equal(plan.lengths, lengths)
    # cognition index = last un-padded token (cogact_arch.py:110-120)
    assert np.array_equal(plan.last_index, lengths - 1)
    assert np.array_equal(plan.kv_end, lengths.astype(np.int32)) and not plan.kv_start.any()


def test_splice_plan_edge_cases():
    from dexbotic_amd.splice import PLAN_PAD, build_splice_plan
    ids = np.array([[5, -200, 7, 8, 0, 0], [9, 10, 11, 12, 13, 14], [1, -200, -200, 2, 3, 0]])
    mask = np.array([[1, 1, 1, 1, 0, 0], [1, 1, 1, 1, 1, 1], [1, 1, 1, 1, 1, 0]], dtype=bool)
    for side in ("right", "left"):
        for max_len in (None, 7):
            p = build_splice_plan(ids, mask, None, 3, max_len, side)
            src, m2, lengths = O.splice_plan(ids, mask, 3, max_len, side)
            assert np.array_equal(p.plan, src), (side, max_len)
            assert np.array_equal(p.attention_mask, m2)
            assert (p.plan[~p.attention_mask] == PLAN_PAD).all()
    # sample 1 has no placeholder but still consumes image block 1; sample 2 uses blocks 2 and 3
    p = build_splice_plan(ids, mask, None, 3)
    assert p.plan[0, 1:4].tolist() == [-1, -2, -3]
    assert p.plan[2, 1:7].tolist() == [-7, -8, -9, -10, -11, -12]
    assert (p.labels == -100).all()
    empty = build_splice_plan(np.zeros((2, 4), dtype=np.int64), np.zeros((2, 4), dtype=bool), None, 3)
    assert empty.plan.shape == (2, 0)


def test_diffusion_tables_match_reference(golden_dir):
    from dexbotic_amd.model.cogact.action_model.diffusion import create_diffusion
    g = np.load(os.path.join(golden_dir, "diffusion_tables.npz"))
    tr = create_diffusion("", "squaredcos_cap_v2", diffusion_steps=100, sigma_small=True, learn_sigma=False)
    assert np.array_equal(tr.betas, g["betas"]) and np.array_equal(tr.alphas_cumprod, g["alphas_cumprod"])
    assert np.array_equal(tr.sqrt_alphas_cumprod, g["sqrt_alphas_cumprod"])
    assert np.array_equal(tr.sqrt_one_minus_alphas_cumprod, g["sqrt_one_minus_alphas_cumprod"])
    for n in (1, 2, 5, 10, 20, 25, 50):
        dd = create_diffusion(f"ddim{n}", "squaredcos_cap_v2", diffusion_steps=100, sigma_small=True, learn_sigma=False)
        assert dd.timestep_map == g[f"ddim{n}/timestep_map"].tolist()
        for k in ("alphas_cumprod", "alphas_cumprod_prev", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod"):
            assert np.array_equal(getattr(dd, k), g[f"ddim{n}/{k}"]), (n, k)
    with pytest.raises(ValueError):
        create_diffusion("ddim30", "squaredcos_cap_v2", diffusion_steps=100, sigma_small=True, learn_sigma=False)


def test_denorm_bit_exact(golden_dir):
    from dexbotic_amd.model.dexbotic_arch import ActionOutputForCausalLM

    class _D(ActionOutputForCausalLM):
        def inference_action(self, *a, **k):
            pass
    g = np.load(os.path.join(golden_dir, "action_bins.npz"))
    den = _D()._denorm(g["normed_all"], {"min": g["mn"].tolist(), "max": g["mx"].tolist()})
    assert np.array_equal(den, g["denorm"])


def test_optimizer_groups_and_schedule(golden_dir):
    from dexbotic_amd.engine import cosine_lr_scale, no_decay_name
    from oracle.gen_golden import no_decay_name as ref_rule
    g, cfg, w = load_golden(golden_dir, "t1")
    from tests.helpers import build_product
    m = build_product(cfg, w, "float32", "cpu", train=True)       # arenas + module tree only; no kernel runs on the host
    for name in w:
        assert no_decay_name(name, m.store) == ref_rule(name), name
    # ... and by module TYPE, the way the reference selects them (base_exp.py:101-102)
    import torch.nn as nn
    from transformers.trainer_pt_utils import get_parameter_names
    decay = [n for n in get_parameter_names(m, [nn.LayerNorm]) if "bias" not in n]
    for name in w:
        assert (name not in decay) == ref_rule(name), name
    assert cosine_lr_scale(0, 100) == 1.0 and abs(cosine_lr_scale(50, 100) - 0.5) < 1e-12
    assert cosine_lr_scale(100, 100) == 0.0 and cosine_lr_scale(5, 100, 10) == 0.5


def test_registry_and_config_roundtrip(tmp_path):
    import dexbotic_amd
    reg = dexbotic_amd.model_registry()
    assert set(reg) >= {"dexbotic", "dexbotic_cogact"}
    c = product_config(CFGS["t1"])
    assert c.model_type == "dexbotic_cogact" and c.hidden_size == 256 and c.vocab_size == 512
    c.save_pretrained(str(tmp_path))
    c2 = type(c).from_pretrained(str(tmp_path))
    assert c2.llm_config.to_dict() == c.llm_config.to_dict()
    assert c2.action_model_type == "DiT-T" and c2.chunk_size == 16
    from dexbotic_amd.model.cogact.action_model.builder import build_action_model
    with pytest.raises(RuntimeError):
        build_action_model(c)                      # one-argument reference signature: needs an open build context
    from dexbotic_amd.engine import ParamStore, building
    with building(ParamStore("cpu")):
        with pytest.raises(ValueError):
            build_action_model(object())
    # HF registry: the reference's model_type strings resolve to the native config classes
    from transformers import AutoConfig
    c3 = AutoConfig.from_pretrained(str(tmp_path))
    assert type(c3) is type(c) and c3.llm_config.to_dict() == c.llm_config.to_dict()


def test_pi0_and_memvla_configs_resolve_through_autoconfig(tmp_path):
    """``AutoConfig.from_pretrained`` on a directory whose config.json carries the reference's model_type strings
    ("dexbotic_pi0": pi0_arch.py:54, "dexbotic_memvla": memvla_arch.py:21) yields the native PretrainedConfig subclasses with
    their nested configs intact"""
    from transformers import AutoConfig, PretrainedConfig
    from dexbotic_amd.model.memvla.memvla_arch import MemVLAConfig
    from dexbotic_amd.model.pi0.pi0_arch import Pi0Config
    gem = dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1, head_dim=32,
               intermediate_size=192, vocab_size=300)
    c = Pi0Config(llm_config=gem, action_config=dict(gem, hidden_size=64, intermediate_size=96),
                  vision_config=dict(hidden_size=64, num_hidden_layers=2), chunk_size=6, compute_dtype="bfloat16")
    assert isinstance(c, PretrainedConfig) and c.hidden_size == 128 and c.vocab_size == 300
    c.save_pretrained(str(tmp_path / "pi0"))
    c2 = AutoConfig.from_pretrained(str(tmp_path / "pi0"))
    assert type(c2) is Pi0Config and c2.to_dict() == c.to_dict()
    assert c2.action_config.hidden_size == 64 and c2.chunk_size == 6 and c2.compute_dtype == "bfloat16"
    m = MemVLAConfig(llm_config=product_config(CFGS["t1"]).llm_config, per_token_size=32, mem_length=4)
    assert m.retrieval_dropout == 0.1                      # the reference hard-codes dropout 0.1 in its retrieval blocks
    m.save_pretrained(str(tmp_path / "mem"))
    m2 = AutoConfig.from_pretrained(str(tmp_path / "mem"))
    assert type(m2) is MemVLAConfig and m2.per_token_size == 32 and m2.mem_length == 4 and m2.retrieval_dropout == 0.1


def test_automodel_from_pretrained_builds_the_native_class(golden_dir, tmp_path):
    """AutoModel.register(Config, ForCausalLM) (the reference's dm0 / pi05 registration pattern): a checkpoint directory written
    by save_pretrained resolves to the native class with identical tensors; construction and load need no GPU"""
    import torch
    from transformers import AutoModel
    from dexbotic_amd.model.cogact.cogact_arch import CogACTForCausalLM
    from tests.helpers import build_product, load_golden
    g, cfg, w = load_golden(golden_dir, "t1")
    m = build_product(cfg, w, "float32", "cpu", train=False)
    m.save_pretrained(str(tmp_path))
    m2 = AutoModel.from_pretrained(str(tmp_path), device="cpu")
    assert type(m2) is CogACTForCausalLM
    sd, sd2 = m.state_dict(), m2.state_dict()
    assert sd.keys() == sd2.keys() and all(torch.equal(sd[k], sd2[k]) for k in sd)


def test_action_norm_and_2string_bit_exact(golden_dir):
    """row A9, encode direction, in the PRODUCT (dexbotic_amd/data/dataset/transform/action.py) against the integer rows the
    reference's ActionNormAnd2String produced (tests/golden/action_bins.npz): normalised values, bins (round-half-even on
    exact halves) and strings bit-exact; plus the episode-dict contract of __call__"""
    import os
    import numpy as np
    from dexbotic_amd.data.dataset.transform.action import ActionNormAnd2String
    g = np.load(os.path.join(golden_dir, "action_bins.npz"))
    V = int(g["vocab"])
    tf = ActionNormAnd2String({"default": {"min": g["mn"].tolist(), "max": g["mx"].tolist()}}, vocab_size=V)
    normed = tf._norm_action(g["action"], g["mn"], g["mx"])
    assert np.array_equal(normed, g["normed"])
    bins = tf._action2bin(g["normed_all"], V)
    assert np.array_equal(bins, g["bins"])
    assert tf._bin2string(bins, tf.string_format) == g["strings"].tolist()
    ep = {"action": g["action"].copy(), "prompt": ["pick"], "meta_data": {"dataset": "unseen"}}
    out = tf(ep)
    assert np.array_equal(out["action"], g["normed"])
    assert out["answer"] == tf._bin2string(tf._action2bin(g["normed"], V), " {value}")
    ep2 = {"action": g["action"].copy(), "prompt": ["pick"], "meta_data": {"dataset": "unseen"}, "answer": ["kept"]}
    assert tf(ep2)["answer"] == ["kept"]
    assert tf({"prompt": ["x"]}) == {"prompt": ["x"]}
    # per-dataset / per-prompt statistics and scalar broadcasting
    tf2 = ActionNormAnd2String({"default": {"min": -1, "max": 1}, "d1": {"default": {"min": [-2.0], "max": [2.0]},
                                                                         "open": {"min": [-4] * 7, "max": [4] * 7}}})
    a = np.full((2, 7), 1.0)
    assert np.allclose(tf2({"action": a.copy(), "prompt": ["zz"], "meta_data": {"dataset": "d1"}})["action"], 0.5)
    assert np.allclose(tf2({"action": a.copy(), "prompt": ["open"], "meta_data": {"dataset": "d1"}})["action"], 0.25)
    assert np.allclose(tf2({"action": a.copy(), "prompt": ["open"], "meta_data": {"dataset": "other"}})["action"], 1.0, atol=1e-7)


def test_uncovered_ranges_of_the_gradient_arena():
    """host bookkeeping of the single-GPU clip norm: [lo, hi) minus the slots a dW epilogue already accounted for, minus the
    slots the path never writes and the frozen ones; neighbours separated only by alignment padding merge"""
    import torch
    from dexbotic_amd.engine import ALIGN, ParamStore
    st = ParamStore("cpu", torch.float32)
    st.new_bucket()
    st.register([("a.ln", (10,))])
    st.register([("a.w", (8, 16)), ("a.b", (16,))])
    st.new_bucket()
    st.register([("b.w", (4, 4))])
    st.register([("b.frozen", (5,))])
    st.register([("b.unused", (7,))])
    st.finalize(train=True)
    st.params["b.frozen"].requires_grad_(False)
    st.set_expected(["b.unused"])
    st.begin_step()
    total = st.total
    off = {n: (s.offset, s.offset + s.numel) for n, s in st.slots.items()}
    # nothing covered: every trainable, expected slot; slots of one group are adjacent, groups are ALIGN apart (padding merges)
    r = st.uncovered_ranges(0, total)
    assert r[0][0] == off["a.ln"][0] and r[-1][1] == off["b.w"][1]
    assert all(b - a > 0 for a, b in r)
    covered_elems = sum(b - a for a, b in r)
    assert covered_elems >= 10 + 8 * 16 + 16 + 16 and covered_elems <= total
    assert not any(a <= off["b.unused"][0] < b for a, b in r) and not any(a <= off["b.frozen"][0] < b for a, b in r)
    # the big weight covered by its product's epilogue: its neighbours remain, split around it
    st._ssq_covered.add("a.w")
    r = st.uncovered_ranges(0, total)
    assert (off["a.ln"][0], off["a.ln"][1]) in r
    assert any(a <= off["a.b"][0] and off["a.b"][1] <= b for a, b in r)       # (merged with b.w across the alignment gap)
    assert not any(a < off["a.w"][1] and off["a.w"][0] < b for a, b in r)
    # a sub-range (one bucket)
    lo, hi = st.bucket_ranges[2]
    r2 = st.uncovered_ranges(lo, hi)
    assert r2 == [off["b.w"]]
    st.begin_step()
    assert not st._ssq_covered
    assert ALIGN == 64


def test_backward_reenters_the_fp32_product_mode_of_its_forward():
    """functional._StoreFn: a Function's backward runs the fp32 head products in the mode its forward ran in (exact fp32 MFMA or
    the split-bf16 product) — an external training loop (HF Trainer) calls loss.backward() after the scope that set the mode for
    the forward has been left"""
    import torch
    from dexbotic_amd import functional as Fn
    from dexbotic_amd import kernels as K
    seen = {}

    class Probe(Fn._StoreFn):
        @staticmethod
        def forward(ctx, x):
            seen["fwd"] = K.F32_GEMM_MODE
            return x * 2.0

        @staticmethod
        def backward(ctx, dy):
            seen["bwd"] = K.F32_GEMM_MODE
            return dy * 2.0

    assert K.F32_GEMM_MODE == "exact"
    x = torch.ones(3, requires_grad=True)
    with K.f32_gemm_mode("bf16x3"):
        y = Probe.apply(x)
    assert K.F32_GEMM_MODE == "exact"
    y.sum().backward()
    assert seen == {"fwd": "bf16x3", "bwd": "bf16x3"} and K.F32_GEMM_MODE == "exact"
    assert torch.equal(x.grad, torch.full((3,), 2.0))
    y2 = Probe.apply(x)                                     # and the other way round: forward exact, backward inside a bf16x3 scope
    with K.f32_gemm_mode("bf16x3"):
        y2.sum().backward()
    assert seen == {"fwd": "exact", "bwd": "exact"}


def test_coalesce_batches_interleaves_draws_pads_ragged_rows_and_refuses_what_it_does_not_know():
    """host arithmetic of trainer.coalesce_batches (the accumulation micro-batches of an optimizer step as ONE batch): rows are
    concatenated, the injected draws ([R * B, ...], row r * B + b as cogact_arch.py:110-125 repeats the batch) re-interleaved per
    repeat, token rows right-padded to the longest micro-batch (id 0 / mask 0 / label -100)"""
    from dexbotic_amd.trainer import coalesce_batches
    B, R, S = 3, 4, 6
    rs = np.random.RandomState(0)

    def mk(S_, off):
        return dict(input_ids=torch.arange(B * S_).reshape(B, S_) + off, attention_mask=torch.ones(B, S_, dtype=torch.bool),
                    labels=torch.arange(B * S_).reshape(B, S_) + off, images=torch.from_numpy(rs.randn(B, 3, 4, 4).astype(np.float32)),
                    actions=torch.from_numpy(rs.randn(B, 16, 7).astype(np.float32)),
                    noise=torch.from_numpy(rs.randn(R * B, 16, 7).astype(np.float32)), timesteps=torch.arange(R * B) + off,
                    drop_ids=torch.from_numpy(rs.rand(R * B) < 0.5))
    a, b, c = mk(S, 0), mk(S + 2, 1000), mk(S, 2000)
    m = coalesce_batches([a, b, c])
    assert m["input_ids"].shape == (3 * B, S + 2) and m["images"].shape[0] == 3 * B and m["noise"].shape[0] == 3 * R * B
    assert torch.equal(m["input_ids"][:B, :S], a["input_ids"]) and torch.equal(m["input_ids"][B:2 * B], b["input_ids"])
    assert (m["input_ids"][:B, S:] == 0).all() and not m["attention_mask"][:B, S:].any() and (m["labels"][2 * B:, S:] == -100).all()
    assert m["attention_mask"].dtype == torch.bool and m["attention_mask"][B:2 * B].all()
    for key in ("noise", "timesteps", "drop_ids"):
        for r in range(R):
            for j, src in enumerate((a, b, c)):
                assert torch.equal(m[key][r * 3 * B + j * B:r * 3 * B + (j + 1) * B], src[key][r * B:(r + 1) * B]), (key, r, j)
    # what the merged batch means to the model: repeat(R) of the merged rows lines up with the merged draws
    acts_rep = m["actions"].repeat(R, 1, 1)
    for r in range(R):
        assert torch.equal(acts_rep[r * 3 * B + B:r * 3 * B + 2 * B], b["actions"])
    assert coalesce_batches([a, dict(c, states=torch.zeros(B, 2))]) is None                    # other key sets
    assert coalesce_batches([a, dict(c, episode=torch.zeros(B))]) is None
    assert coalesce_batches([dict(a, episode=torch.zeros(B)), dict(c, episode=torch.zeros(B))]) is None      # a key it does not know
    small = {k: v[:2] if k in ("input_ids", "attention_mask", "labels", "images", "actions") else v[:2 * R] for k, v in c.items()}
    assert coalesce_batches([a, small]) is None                                                # unequal micro-batch sizes
    assert coalesce_batches([a, dict(c, images=torch.zeros(B, 3, 8, 8))]) is None              # unequal image shapes
    # no attention mask = every position valid (splice.py): rows of unequal length cannot be padded without turning pad ids
    # into real tokens (advisor, round 4); equal lengths still merge
    nomask = lambda d: {k: v for k, v in d.items() if k != "attention_mask"}
    assert coalesce_batches([nomask(a), nomask(b)]) is None
    assert coalesce_batches([nomask(a), nomask(c)])["input_ids"].shape == (2 * B, S)


def test_exp_config_gradient_checkpointing_is_forwarded_only_on_request(monkeypatch, tmp_path):
    """TrainerConfig.gradient_checkpointing defaults to True in the reference (base_exp.py:245, there to fit 80 GB parts): the
    native trainer keeps activations resident unless DEXBOTIC_AMD_GRAD_CHECKPOINTING=1 forwards the switch to HF's loop, which
    then calls model.gradient_checkpointing_enable() -> ParamStore.recompute"""
    from dexbotic_amd.exp.config import ExpConfig, OptimizerConfig, TrainerConfig
    from dexbotic_amd.exp.trainer import link_exp_config
    exp = ExpConfig(TrainerConfig(output_dir=str(tmp_path), bf16=False), OptimizerConfig())
    assert exp.trainer_config.gradient_checkpointing
    monkeypatch.delenv("DEXBOTIC_AMD_GRAD_CHECKPOINTING", raising=False)
    assert not link_exp_config(exp, report_to=[], use_cpu=True).gradient_checkpointing
    monkeypatch.setenv("DEXBOTIC_AMD_GRAD_CHECKPOINTING", "1")
    args = link_exp_config(exp, report_to=[], use_cpu=True)
    assert args.gradient_checkpointing and args.gradient_checkpointing_kwargs == {"use_reentrant": False}
    exp.trainer_config.gradient_checkpointing = False
    assert not link_exp_config(exp, report_to=[], use_cpu=True).gradient_checkpointing


def test_coalescing_control_flow_holds_merges_and_falls_back():
    """NativeTrainer.micro_step with coalesce on (host logic only; the pass itself is stubbed): the first grad_accum - 1 calls of a
    group hold their batch and return 0, the last runs ONE pass over the merged batch with loss scale 1 (n x the caller's scale
    when given) and returns n x its loss; a group that cannot be merged runs pass by pass and returns the losses' sum; an
    out-of-memory error in the merged pass switches coalescing off and re-runs the group pass by pass"""
    from dexbotic_amd.trainer import NativeTrainer

    dropped = []

    class Store:
        device = torch.device("cpu")
        _accum_stash = {}

        def flush_wgrads(self):
            raise AssertionError("the out-of-memory fallback must DROP the aborted pass's pending products, not launch them")

        def drop_pending_wgrads(self):
            dropped.append(1)
    calls = []

    def mk(fail_merged=False):
        tr = NativeTrainer.__new__(NativeTrainer)
        tr.coalesce, tr.grad_accum, tr._held, tr.coalesced_steps, tr.store = True, 2, [], 0, Store()
        tr.reducer = tr.norm_tracker = tr.last_output = None
        tr.update_due, tr.micro = False, 0

        def _micro(batch, loss_scale=None, group=None):
            if fail_merged and group is not None:
                raise torch.OutOfMemoryError("merged pass does not fit")
            calls.append((batch["input_ids"].shape[0], loss_scale, group))
            return torch.tensor(float(batch["input_ids"].shape[0]))
        tr._micro = _micro
        return tr
    b = lambda n, **extra: dict(input_ids=torch.zeros(n, 4, dtype=torch.long), actions=torch.zeros(n, 2), **extra)
    tr = mk()
    assert tr.micro_step(b(8)).item() == 0.0 and not calls
    out = tr.micro_step(b(8))
    assert calls == [(16, 1.0, 2)] and out.item() == 32.0 and tr.coalesced_steps == 1 and not tr._held
    calls.clear()
    tr.micro_step(b(8), loss_scale=1.0)                                       # HF's summed micro-batch gradients
    tr.micro_step(b(8), loss_scale=1.0)
    assert calls == [(16, 2.0, 2)]
    calls.clear()
    tr.micro_step(b(8))
    out = tr.micro_step(b(8, episode=torch.zeros(8)))                         # a key the merge does not know: pass by pass
    assert calls == [(8, None, None), (8, None, None)] and out.item() == 16.0 and tr.coalesced_steps == 2
    calls.clear()
    tr = mk(fail_merged=True)
    tr.micro_step(b(8))
    out = tr.micro_step(b(8))
    assert calls == [(8, None, None), (8, None, None)] and out.item() == 16.0 and tr.coalesce is False and dropped == [1]
    calls.clear()
    tr.micro_step(b(8))                                                       # ... and stays pass by pass
    assert calls == [(8, None, None)]
    # under data parallelism the merged pass has NO out-of-memory fallback (an aborted backward may have fired collectives)
    calls.clear()
    tr = mk(fail_merged=True)
    tr.reducer = type("R", (), {"world": 2})()
    tr.micro_step(b(8))
    try:
        tr.micro_step(b(8))
        raise AssertionError("expected the out-of-memory error to propagate")
    except torch.OutOfMemoryError:
        pass
    # a SHORT accumulation group (HF 4.51 closes the last group of an epoch without announcing its size): the optimizer facade
    # asks for the step with one of three micro-batches held -> it runs as a group of one at the NOMINAL 1/3 scale
    calls.clear()
    tr = mk()
    tr.grad_accum = 3
    tr.micro_step(b(8))
    assert not calls and len(tr._held) == 1
    tr.close_short_group()
    assert calls == [(8, 1.0 / 3, 1)] and not tr._held and tr.grad_accum == 3 and tr.micro == 0
    calls.clear()
    tr.micro_step(b(8))
    tr.micro_step(b(8))
    tr.close_short_group()                                                    # two of three: merged, scale 2/3 on the merged mean
    assert len(calls) == 1 and calls[0][0] == 16 and abs(calls[0][1] - 2.0 / 3) < 1e-12 and calls[0][2] == 2


def test_gemm_profile_stride_samples_every_product_of_a_layer_equally():
    """bench.py's live roofline times every 5th launch of a layout: with the four products a transformer layer launches per
    layout (qkv, o, gate_up, down) a stride coprime with 4 visits each of them equally often; ``launches`` counts them all"""
    from dexbotic_amd import kernels as K
    key = (0, 1, 1)
    prof = K.GemmProfile(key, stride=5)
    timed = [i % 4 for i in range(28 * 4 * 5) if prof.wants(*key)]
    assert prof.launches(key) == 28 * 4 * 5 and len(timed) == 28 * 4
    assert [timed.count(r) for r in range(4)] == [28] * 4
    assert not prof.wants(2, 1, 1) and prof.launches() == 28 * 4 * 5
    every = K.GemmProfile(key)
    assert all(every.wants(*key) for _ in range(7)) and every.launches(key) == 7


def test_host_thread_pool_is_held_inside_the_cgroup_cpu_quota(tmp_path, monkeypatch):
    """hostcpu: cpu.max "1600000 100000" = 16 CPUs per period (what the MI355X pods have); the pool is clamped below it, never
    raised, and DXA_HOST_THREADS overrides (profiles/r06_process_frame_tail.txt: the /process_frame p90 of 55 ms was this)"""
    import torch
    from dexbotic_amd import hostcpu as H
    (tmp_path / "cpu.max").write_text("1600000 100000\n")
    assert H.cpu_quota(str(tmp_path)) == 16.0
    (tmp_path / "cpu.max").write_text("max 100000\n")
    assert H.cpu_quota(str(tmp_path)) is None
    (tmp_path / "cpu.max").unlink()
    (tmp_path / "cpu").mkdir()
    (tmp_path / "cpu" / "cpu.cfs_quota_us").write_text("250000\n")
    (tmp_path / "cpu" / "cpu.cfs_period_us").write_text("100000\n")
    assert H.cpu_quota(str(tmp_path)) == 2.5
    before = torch.get_num_threads()
    try:
        monkeypatch.setattr(H, "usable_cpus", lambda: 16)
        monkeypatch.delenv("DXA_HOST_THREADS", raising=False)
        monkeypatch.delenv("LOCAL_WORLD_SIZE", raising=False)
        torch.set_num_threads(min(before, 4))
        assert H.limit_host_threads() == min(before, 4)                 # never raised
        monkeypatch.setattr(H, "usable_cpus", lambda: 3)
        assert H.limit_host_threads() == 1                               # 3 - reserve 2
        monkeypatch.setattr(H, "usable_cpus", lambda: 64)
        monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
        torch.set_num_threads(before)
        assert H.limit_host_threads(cap=5) == min(before, 5, 64 // 8 - 2)
        monkeypatch.setenv("DXA_HOST_THREADS", "2")
        assert H.limit_host_threads() == 2
        monkeypatch.setenv("DXA_HOST_THREADS", "0")
        torch.set_num_threads(3)
        assert H.limit_host_threads() == 3                               # 0 = leave torch alone
    finally:
        torch.set_num_threads(before)
