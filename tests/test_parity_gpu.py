"""Model-level parity on the MI355X: the product (dexbotic_amd, HIP kernels through the C ABI) against
  (1) the golden vectors produced by the live reference (tests/golden, oracle/gen_golden.py) and
  (2) the CPU oracle on fresh seeded inputs (shapes the fixtures do not cover).
Bar (BASELINE.json north_star): 7-DoF action chunks / losses / activations within 1e-3 relative in fp32;
bf16 compute is held to a stated looser tolerance against the same fp32 reference."""
import numpy as np
import pytest
import torch

from oracle import cogact_oracle as O
from tests.helpers import assert_chunk_close, build_product, load_golden, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"
FP32_TOL = 1e-3          # north-star tolerance
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _batch(g):
    drop = T(g["drop_u"]) < 0.1
    return dict(input_ids=T(g["input_ids"]), attention_mask=T(g["attention_mask"]), images=T(g["images"]),
                actions=T(g["actions"]), labels=T(g["input_ids"]), noise=T(g["noise"]), timesteps=T(g["timesteps"]),
                drop_ids=drop)


@pytest.mark.parametrize("tag", ["t1", "t2"])
def test_fp32_forward_backward_step_match_reference(golden_dir, tag):
    from dexbotic_amd.engine import OptimConfig
    from dexbotic_amd.trainer import NativeTrainer
    g, cfg, w = load_golden(golden_dir, tag)
    m = build_product(cfg, w, "float32", DEV, train=True)
    m.train()
    tr = NativeTrainer(m, OptimConfig(base_lr=1e-3, weight_decay=0.01, max_grad_norm=1.0))
    m.store.begin_step()
    out = m(**_batch(g))
    assert np.array_equal(m.model._last_plan.attention_mask, g["new_attention_mask"])
    assert rel_err(out.logits.detach().float().cpu().numpy(), g["logits"]) < FP32_TOL
    assert abs(out.loss.item() - float(g["loss"])) < FP32_TOL * abs(float(g["loss"]))
    out.loss.backward()
    st = m.store
    assert sorted(st.never_written()) == sorted(g["no_grad_params"].tolist())
    for key in g.files:
        if key.startswith("grad/"):
            n = key[5:]
            assert rel_err(st.g(n).cpu().numpy(), g[key]) < FP32_TOL, n
        elif key.startswith("gradS/"):
            n = key[6:]
            assert rel_err(st.g(n).reshape(-1)[::97].cpu().numpy(), g[key]) < FP32_TOL, n
            assert abs(st.g(n).double().norm().item() - float(g["gradN/" + n])) < FP32_TOL * float(g["gradN/" + n]), n
    # one optimizer step through the trainer (fresh backward inside), then compare parameters + 2nd-step loss
    loss1 = tr.step(_batch(g))
    assert abs(loss1.item() - float(g["loss"])) < FP32_TOL * abs(float(g["loss"]))
    assert abs(tr.opt.norm.item() - float(g["grad_norm"])) < FP32_TOL * float(g["grad_norm"])
    sd = m.state_dict()
    for key in g.files:
        if key.startswith("param1/"):
            n = key[7:]
            big = np.abs(g["grad/" + n]) > 1e-5       # Adam's first step is sign-like: skip ~0 gradients
            assert np.abs(sd[n].cpu().numpy() - g[key])[big].max() < 5e-6, n
    m.store.begin_step()
    loss2 = m(**_batch(g)).loss.item()
    assert abs(loss2 - float(g["loss_step2"])) < 5e-3 * abs(float(g["loss_step2"]))


@pytest.mark.parametrize("tag", ["t1", "t2"])
def test_fp32_inference_action_matches_reference(golden_dir, tag):
    g, cfg, w = load_golden(golden_dir, tag)
    m = build_product(cfg, w, "float32", DEV, train=False)
    m.eval()
    norms = {"min": g["norm_min"].tolist(), "max": g["norm_max"].tolist()}
    acts, samples, traj = m.inference_action(T(g["infer_ids"]), T(g["images"][:1]),
                                             {"cfg_scale": 1.5, "num_ddim_steps": 10, "action_norms": norms},
                                             noise=T(g["init_noise"]), return_trajectory=True)
    for i in range(len(traj)):
        assert rel_err(traj[i].cpu().numpy(), g["ddim_traj"][i]) < FP32_TOL, i
    assert rel_err(np.array(acts), g["infer_actions"]) < FP32_TOL
    assert_chunk_close(np.array(acts), g["infer_actions"])
    assert isinstance(acts, list) and len(acts) == cfg.chunk_size and len(acts[0]) == cfg.action_dim


@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
def test_inference_action_graph_replay_equals_eager(golden_dir, dtype):
    """the HIP-graph path (captured on the 2nd request of a shape, replayed afterwards) runs the same kernels as
    the eager path: identical actions, also for new inputs fed through the static buffers"""
    g, cfg, w = load_golden(golden_dir, "t1")
    m = build_product(cfg, w, dtype, DEV, train=False)
    m.eval()
    norms = {"min": g["norm_min"].tolist(), "max": g["norm_max"].tolist()}
    args = {"cfg_scale": 1.5, "num_ddim_steps": 10, "action_norms": norms}
    ids, img, noise = T(g["infer_ids"]), T(g["images"][:1]), T(g["init_noise"])
    eager = m.inference_action(ids, img, dict(args, use_graph=False), noise=noise)
    if dtype == "float32":
        assert rel_err(np.array(eager), g["infer_actions"]) < FP32_TOL
        assert_chunk_close(np.array(eager), g["infer_actions"])
    for call in range(4):                                   # 1: eager warm-up, 2: capture + replay, 3-4: replay
        acts = m.inference_action(ids, img, dict(args, use_graph=True), noise=noise)
        assert acts == eager, call
    assert next(iter(m._infer_graphs.values()))["graph"] is not None
    ids2 = ids.clone()
    ids2[0, -1] = (ids2[0, -1] + 1) % cfg.vocab_size        # same shape, different prompt / image / noise
    img2, noise2 = img * 0.5, noise * 0.9
    want = m.inference_action(ids2, img2, dict(args, use_graph=False), noise=noise2)
    assert m.inference_action(ids2, img2, dict(args, use_graph=True), noise=noise2) == want
    assert want != eager


@pytest.mark.parametrize("tag", ["t1", "t2"])
def test_bf16_tracks_fp32_reference(golden_dir, tag):
    """bf16 MFMA path (flash attention, bf16 GEMMs, fp32 head): stated tolerance 3e-2 on hidden states,
    2e-2 on the loss, against the fp32 reference outputs."""
    g, cfg, w = load_golden(golden_dir, tag)
    m = build_product(cfg, w, "bfloat16", DEV, train=True)
    m.train()
    m.store.begin_step()
    out = m(**_batch(g))
    assert rel_err(out.logits.detach().float().cpu().numpy(), g["logits"]) < 3e-2
    assert abs(out.loss.item() - float(g["loss"])) < 2e-2 * abs(float(g["loss"]))
    out.loss.backward()
    st = m.store
    for n in ("model.llm.layers.0.self_attn.q_proj.weight", "model.mm_projector.0.weight",
              "model.action_head.net.blocks.1.mlp.fc1.weight"):
        gn = float(g["gradN/" + n])
        assert abs(st.g(n).double().norm().item() - gn) < 6e-2 * gn, n


def test_fp32_matches_oracle_on_fresh_inputs_with_left_padding_and_views():
    """shapes/flags the fixtures do not cover: left padding, 2 views + padding, no CFG drop, R=2."""
    from oracle.weights import cogact_shapes, make_weights
    cfg = O.OracleConfig(vocab_size=300, hidden_size=256, intermediate_size=384, num_hidden_layers=2,
                         num_attention_heads=2, num_key_value_heads=2, v_hidden=128, v_inter=192, v_layers=3,
                         v_heads=2, dit_hidden=128, dit_depth=2, dit_heads=2, tokenizer_padding_side="left")
    w = make_weights(cogact_shapes(cfg), 77)
    rs = np.random.RandomState(5)
    B, L, V, R = 3, 10, 2, 2
    ids = rs.randint(5, 290, size=(B, L)).astype(np.int64)
    ids[:, 2] = -200
    mask = np.ones((B, L), dtype=bool)
    mask[1, 7:] = False
    mask[2, 9:] = False
    images = np.clip(rs.standard_normal((B, V, 3, 56, 56)), -2.5, 2.5).astype(np.float32)
    actions = rs.uniform(-1, 1, size=(B, 112)).astype(np.float32)
    noise = rs.standard_normal((R * B, 16, 7)).astype(np.float32)
    ts = rs.randint(0, 100, size=(R * B,)).astype(np.int64)
    sd = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in w.items()}
    ref = O.cogact_forward(sd, cfg, torch.from_numpy(ids), torch.from_numpy(mask), torch.from_numpy(images),
                           torch.from_numpy(actions), torch.from_numpy(noise), torch.from_numpy(ts), None, R)
    ref["loss"].backward()
    m = build_product(cfg, w, "float32", DEV, train=True)
    m.config.tokenizer_padding_side = "left"
    m.eval()                              # no CFG token drop
    m.store.begin_step()
    out = m(input_ids=T(ids), attention_mask=T(mask), images=T(images), actions=T(actions), noise=T(noise),
            timesteps=T(ts), repeated_diffusion_steps=R)
    valid = ref["attention_mask"].numpy()
    got = out.logits.detach().float().cpu().numpy()
    assert rel_err(got[valid], ref["logits"].detach().numpy()[valid]) < FP32_TOL
    assert abs(out.loss.item() - ref["loss"].item()) < FP32_TOL * ref["loss"].item()
    out.loss.backward()
    for n in ("model.llm.layers.1.mlp.down_proj.weight", "model.mm_projector.2.weight",
              "model.mm_vision_tower.vision_tower.embeddings.position_embedding.weight",
              "model.llm.embed_tokens.weight", "model.action_head.net.z_embedder.linear.weight"):
        assert rel_err(m.store.g(n).cpu().numpy(), sd[n].grad.numpy()) < FP32_TOL, n


def test_gradient_accumulation_equals_big_batch(golden_dir):
    from dexbotic_amd.engine import OptimConfig
    from dexbotic_amd.trainer import NativeTrainer
    g, cfg, w = load_golden(golden_dir, "t1")
    b = _batch(g)
    m1 = build_product(cfg, w, "float32", DEV, train=True)
    m1.store.begin_step()
    m1(**b).loss.backward()
    g_full = m1.store.grad.clone()
    m2 = build_product(cfg, w, "float32", DEV, train=True)
    tr = NativeTrainer(m2, OptimConfig(base_lr=0.0), grad_accum=3)
    R, B = 4, 3
    for i in range(B):                       # one sample per micro-batch, same injected draws
        sel = torch.arange(R, device=DEV) * B + i
        mb = dict(input_ids=b["input_ids"][i:i + 1], attention_mask=b["attention_mask"][i:i + 1],
                  images=b["images"][i:i + 1], actions=b["actions"][i:i + 1], noise=b["noise"][sel],
                  timesteps=b["timesteps"][sel], drop_ids=b["drop_ids"][sel])
        tr.step(mb)
    assert rel_err(m2.store.grad.cpu().numpy(), g_full.cpu().numpy()) < 1e-4
    # the global norm under accumulation: the LAST micro-batch's dW epilogues square what they store (= the accumulated
    # gradient), the rest is read back — together exactly sum(g^2) of the arena
    want = float((m2.store.grad.double() ** 2).sum())
    assert m2.store.epi_sumsq and m2.store._ssq_cursor > 0, "the epilogue path was not taken on the last micro-batch"
    assert abs(float(tr._sumsq) - want) < 1e-5 * want, (float(tr._sumsq), want)


@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
@pytest.mark.parametrize("cfg_scale", [1.5, 1.0])
def test_whole_sampler_in_one_launch_equals_the_per_step_sampler(golden_dir, dtype, cfg_scale, monkeypatch):
    """dxa_dit_sample_fwd (all DDIM steps — embedders, blocks, final layer, guidance, update — in ONE persistent launch) against
    the per-step path (DXA_DIT_SAMPLER=0: one persistent block launch + ~14 small launches per step) and, in fp32 with
    guidance, against the reference's golden action chunk; three requests in a row (the barrier / tile counters are left clean)"""
    g, cfg, w = load_golden(golden_dir, "t2")            # t2: hidden 192 x 3 heads x 64, 3 DiT blocks: the fused kernels' shapes
    m = build_product(cfg, w, dtype, DEV, train=False)
    m.eval()
    with torch.no_grad():
        assert m.model.action_head.net.fused_sampler_ok(2 if cfg_scale > 1.0 else 1, cfg.chunk_size + 1)
    norms = {"min": g["norm_min"].tolist(), "max": g["norm_max"].tolist()}
    args = {"cfg_scale": cfg_scale, "num_ddim_steps": 10, "action_norms": norms}
    ids, img, noise = T(g["infer_ids"]), T(g["images"][:1]), T(g["init_noise"])
    got = [np.asarray(m.inference_action(ids, img, args, noise=noise)) for _ in range(3)]
    assert m.model.action_head.net.used_fused
    assert np.array_equal(got[0], got[1]) and np.array_equal(got[0], got[2])
    exact = got[0]
    if dtype == "bfloat16":
        # a model served in bfloat16 samples with bf16 MFMA operands (dit_sample_bf16_k, like the reference's bf16 head):
        # DXA_DIT_BF16=0 is the exact-fp32 one-launch sampler, which is what equals the per-step path to rounding
        assert m.model.action_head.net._bf16_sampler(2 if cfg_scale > 1.0 else 1, cfg.chunk_size + 1)
        monkeypatch.setenv("DXA_DIT_BF16", "0")
        exact = np.asarray(m.inference_action(ids, img, args, noise=noise))
        assert m.model.action_head.net.used_fused
        # operand rounding over 10 steps x 3 blocks of this toy head, whose random weights amplify it (measured 3.2e-2 of the largest
        # de-normalised action; a DiT-B-size random head: 2.6e-3, bounded in tests/test_kernels_gpu.py::test_dit_sample_bf16_operands)
        assert rel_err(got[0], exact) < 5e-2, rel_err(got[0], exact)          # (1.5 x the 3.2e-2 measured)
    monkeypatch.setenv("DXA_DIT_SAMPLER", "0")
    want = np.asarray(m.inference_action(ids, img, args, noise=noise))
    assert rel_err(exact, want) < 2e-4, rel_err(exact, want)
    if dtype == "float32" and cfg_scale == 1.5:
        assert rel_err(got[0], g["infer_actions"]) < FP32_TOL
        assert_chunk_close(got[0], g["infer_actions"])


def test_two_micro_batches_write_each_weight_gradient_once(golden_dir, monkeypatch):
    """gradient accumulation over exactly two micro-batches (the reference recipe, cogact_exp.py:41-46): the linears' dW is ONE
    product over both micro-batches' (dY, X) pairs (ParamStore.accum_merge).  bf16 model: gradients, global norm and the
    parameters after the update against the read-modify-write form (DXA_NO_ACCUM_MERGE) — two fp32 accumulation chains added
    vs one chain: equal to fp32 rounding."""
    from dexbotic_amd.engine import OptimConfig
    from dexbotic_amd.trainer import NativeTrainer
    g, cfg, w = load_golden(golden_dir, "t2")
    b = _batch(g)
    B = b["input_ids"].shape[0]
    R = b["noise"].shape[0] // B
    res = {}
    for tag in ("merge", "rmw"):
        if tag == "rmw":
            monkeypatch.setenv("DXA_NO_ACCUM_MERGE", "1")
        m = build_product(cfg, w, "bfloat16", DEV, train=True)
        tr = NativeTrainer(m, OptimConfig(base_lr=1e-3), grad_accum=2)
        half = B // 2
        for i in range(2):
            rows = torch.arange(i * half, (i + 1) * half, device=DEV)
            sel = (torch.arange(R, device=DEV)[:, None] * B + rows[None, :]).reshape(-1)
            mb = dict(input_ids=b["input_ids"][i * half:(i + 1) * half], attention_mask=b["attention_mask"][i * half:(i + 1) * half],
                      images=b["images"][i * half:(i + 1) * half], actions=b["actions"][i * half:(i + 1) * half], noise=b["noise"][sel],
                      timesteps=b["timesteps"][sel], drop_ids=b["drop_ids"][sel])
            tr.micro_step(mb)
        assert m.store.accum_merge == (tag == "merge") and not m.store._accum_stash
        res[tag] = (m.store.grad.clone(), float(tr._sumsq))
        tr.apply_update()
        torch.cuda.synchronize()
        res[tag] += (m.store.master.clone(),)
    gm, gr = res["merge"][0], res["rmw"][0]
    assert rel_err(gm.cpu().numpy(), gr.cpu().numpy()) < 1e-5
    assert abs(res["merge"][1] - res["rmw"][1]) < 1e-5 * res["rmw"][1]
    assert rel_err(res["merge"][2].cpu().numpy(), res["rmw"][2].cpu().numpy()) < 1e-6


@pytest.mark.parametrize("mode", ["3", "1"])
def test_gradient_work_on_the_side_stream_is_bit_identical(golden_dir, monkeypatch, mode):
    """the fp32 head's dW products and the bias column sums run on a side HIP stream beside the dX chain (trainer.NativeTrainer,
    DXA_WGRAD_STREAM, default 3; 1 = every dW product): same kernels on the same operands, joined before anything reads the
    gradient arena — gradients, the clipped norm and the parameters after two optimizer steps (accumulation 2 on the second model
    pass: the merged two-segment dW products) equal the single-stream run bit for bit"""
    from dexbotic_amd.engine import OptimConfig
    from dexbotic_amd.trainer import NativeTrainer
    g, cfg, w = load_golden(golden_dir, "t2")
    b = _batch(g)
    B = b["input_ids"].shape[0]
    R = b["noise"].shape[0] // B
    half = B // 2
    micro = []
    for i in range(2):
        rows = torch.arange(i * half, (i + 1) * half, device=DEV)
        sel = (torch.arange(R, device=DEV)[:, None] * B + rows[None, :]).reshape(-1)
        micro.append(dict(input_ids=b["input_ids"][i * half:(i + 1) * half], attention_mask=b["attention_mask"][i * half:(i + 1) * half],
                          images=b["images"][i * half:(i + 1) * half], actions=b["actions"][i * half:(i + 1) * half],
                          noise=b["noise"][sel], timesteps=b["timesteps"][sel], drop_ids=b["drop_ids"][sel]))
    res = {}
    for m_ in ("0", mode):
        monkeypatch.setenv("DXA_WGRAD_STREAM", m_)
        out = []
        for dtype in ("float32", "bfloat16"):
            m = build_product(cfg, w, dtype, DEV, train=True)
            tr = NativeTrainer(m, OptimConfig(base_lr=1e-3, weight_decay=0.01, max_grad_norm=1.0))
            assert (m.store.wgrad_stream is not None) == (m_ != "0")
            tr.step(b)
            g1 = m.store.grad.clone()
            tr.set_grad_accum(2)
            for mb in micro:
                tr.step(mb)
            torch.cuda.synchronize()
            out.append((g1, m.store.grad.clone(), float(tr.opt.norm.item()), m.store.master.clone()))
        res[m_] = out
    for a, c in zip(res["0"], res[mode]):
        assert torch.equal(a[0], c[0]) and torch.equal(a[1], c[1]) and a[2] == c[2] and torch.equal(a[3], c[3])
        assert a[0].abs().max().item() > 0


def test_adamw_leaves_untouched_embedding_chunks_alone_and_nothing_changes(golden_dir, monkeypatch):
    """dxa_adamw_desc.chunk_state: chunks of embed_tokens.weight that never saw a non-zero gradient are skipped while their gradient
    is all-zero (weight decay 0) — what torch's AdamW computes for them is p, m, v unchanged.  Three optimizer steps with OTHER token
    ids each time, against DXA_ADAMW_SPARSE=0: parameters and both moments bit-identical; most embedding chunks stay in state 1;
    with weight decay on, nothing is skipped and the results are again bit-identical"""
    from dexbotic_amd.engine import OptimConfig
    from dexbotic_amd.trainer import NativeTrainer
    g, cfg, w = load_golden(golden_dir, "t2")
    for wd in (0.0, 0.01):
        res = {}
        for sparse in ("1", "0"):
            monkeypatch.setenv("DXA_ADAMW_SPARSE", sparse)
            m = build_product(cfg, w, "bfloat16", DEV, train=True)
            # (chunks of 128 elements: a couple of toy-width table rows each, so that three steps leave table chunks untouched)
            tr = NativeTrainer(m, OptimConfig(base_lr=1e-3, weight_decay=wd, max_grad_norm=1.0, chunk=128))
            assert (tr.opt.chunk_state is not None) == (sparse == "1")
            states = []
            for k in range(3):
                b = _batch(g)
                ids = b["input_ids"].clone()
                text = ids >= 0
                ids[text] = (ids[text] + 7 * k) % cfg.vocab_size          # other instruction tokens every step
                b["input_ids"], b["labels"] = ids, ids
                tr.step(b)
                if tr.opt.chunk_state is not None:
                    states.append(tr.opt.chunk_state.clone())
            torch.cuda.synchronize()
            res[sparse] = (m.store.master.clone(), tr.opt.m.clone(), tr.opt.v.clone(), m.store.shadow.clone(), states)
        for i in range(4):
            assert torch.equal(res["1"][i], res["0"][i]), (wd, i)
        st = res["1"][4]
        cand = st[0] > 0
        assert cand.any()
        if wd == 0.0:
            # activated chunks only ever grow, and some table chunks are still untouched after three steps
            assert ((st[0] == 2) & ~(st[1] == 2)).sum().item() == 0 and ((st[1] == 2) & ~(st[2] == 2)).sum().item() == 0
            assert (st[0] == 2).any() and (st[2][cand] == 1).any()
        else:
            assert (st[2][cand] == 1).all()                                  # decayed every step: never skipped, never promoted


def test_graph_capture_holds_the_garbage_collector_off():
    """graphs.capture: Python's cyclic collector must not run while a stream is capturing (destructors of older models / trainers /
    graphs that talk to the HIP runtime abort the process in global capture mode — seen once in this suite, round 6) and is
    switched back on afterwards, also when the captured body raises"""
    import gc

    from dexbotic_amd import graphs
    s = torch.cuda.Stream()
    x = torch.zeros(8, device=DEV)
    assert gc.isenabled()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with graphs.capture(g, s):
            assert not gc.isenabled()
            y = x + 1
    assert gc.isenabled()
    g.replay()
    torch.cuda.synchronize()
    assert float(y.sum()) == 8.0
    g2 = torch.cuda.CUDAGraph()
    with pytest.raises(RuntimeError, match="boom"):
        with torch.cuda.stream(s):
            with graphs.capture(g2, s):
                raise RuntimeError("boom")
    assert gc.isenabled()
