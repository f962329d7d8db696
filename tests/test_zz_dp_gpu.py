"""Training-step reproducibility and the data-parallel path on ONE MI355X (sorts last on purpose: a failure here must
not keep `pytest -x` from reaching the model rows).

  * the plain fp32 step is bitwise reproducible from identical state (every reduction in the step has a fixed order:
    two-stage column sums, split-K gathers in slice order, bucket-ordered grad-norm folds);
  * the RCCL reducer path at world size 1 (reducer forced on) reproduces the plain step — fp32 communication bit for
    bit, bf16 communication within the rounding of the bf16 gradient copy;
  * MemVLA (parameters applied once PER SAMPLE) through NativeTrainer: the folded grad-norm equals the norm of the
    final gradient arena and the reference's golden grad-norm.
Reference behaviour: DDP's deterministic bucketed mean, dexbotic/exp/trainer.py:110,121."""
import os

import numpy as np
import pytest
import torch

from tests.helpers import build_product, load_golden, rel_err
from tests.test_parity_gpu import _batch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _trainer(m, clip=1.0, **kw):
    from dexbotic_amd.engine import OptimConfig
    from dexbotic_amd.trainer import NativeTrainer
    return NativeTrainer(m, OptimConfig(base_lr=1e-3, weight_decay=0.01, max_grad_norm=clip), **kw)


def test_plain_fp32_step_is_bitwise_reproducible(golden_dir):
    g, cfg, w = load_golden(golden_dir, "t1")
    ref = None
    for rep in range(10):
        m = build_product(cfg, w, "float32", DEV, train=True)
        m.train()
        tr = _trainer(m, min_bucket_bytes=1 << 14)
        for _ in range(2):
            tr.step(_batch(g))
        torch.cuda.synchronize()
        cur = (m.store.master.clone(), m.store.grad.clone(), tr.opt.norm.clone())
        if ref is None:
            ref = cur
            continue
        assert torch.equal(cur[2], ref[2]), f"grad norm differs on repetition {rep}: {cur[2].item()} vs {ref[2].item()}"
        assert torch.equal(cur[1], ref[1]), f"gradient arena differs on repetition {rep}"
        assert torch.equal(cur[0], ref[0]), f"master arena differs on repetition {rep}"


def test_folded_grad_norm_equals_arena_norm(golden_dir):
    """the bucket-by-bucket sum of squares (side stream, under the backward) must see FINAL gradients, the embedding
    slice included (its scatter-add is the last kernel of the backward)"""
    g, cfg, w = load_golden(golden_dir, "t1")
    m = build_product(cfg, w, "float32", DEV, train=True)
    m.train()
    tr = _trainer(m, min_bucket_bytes=1 << 12)
    tr.step(_batch(g))
    torch.cuda.synchronize()
    full = m.store.grad.double().norm().item()
    assert abs(tr.opt.norm.item() - full) <= 1e-6 * full
    emb = m.store.g(m.model.llm.embed_name)
    assert emb.abs().sum().item() > 0
    assert abs(tr.opt.norm.item() - float(g["grad_norm"])) < 1e-3 * float(g["grad_norm"])


@pytest.mark.parametrize("algo", ["rs_ag", "allreduce"])
def test_rccl_reducer_path_equals_plain_step(golden_dir, algo):
    import torch.distributed as dist
    g, cfg, w = load_golden(golden_dir, "t1")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        res = {}
        for tag, force, comm in (("plain", False, torch.float32), ("plain_epi", False, torch.float32),
                                 ("fp32", True, torch.float32), ("bf16", True, torch.bfloat16)):
            m = build_product(cfg, w, "float32", DEV, train=True)
            m.train()
            tr = _trainer(m, force_reducer=force, grad_comm_dtype=comm, min_bucket_bytes=1 << 16, grad_sync=algo)
            assert (tr.reducer is not None) == force
            # one GPU without a reducer takes sum(g^2) from the dW epilogues (another summation order): "plain" is the
            # read-back pass the reducer path also uses, "plain_epi" the default
            assert m.store.epi_sumsq == (not force)
            if tag == "plain":
                m.store.epi_sumsq = False
            losses = [tr.step(_batch(g)).item() for _ in range(2)]
            torch.cuda.synchronize()
            res[tag] = (losses, tr.opt.norm.clone(), m.store.master.clone())
            if force:
                assert tr.reducer.bytes_reduced > 0
        # fp32 communication at world size 1 is the identity: the whole step is bit-identical to the plain one
        assert res["fp32"][0] == res["plain"][0]
        assert torch.equal(res["fp32"][1], res["plain"][1])
        assert torch.equal(res["fp32"][2], res["plain"][2])
        # the epilogue sums give the same norm up to fp32 summation order, hence the same step up to that
        assert abs(res["plain_epi"][1].item() - res["plain"][1].item()) <= 2e-6 * res["plain"][1].item()
        assert res["plain_epi"][0][0] == res["plain"][0][0]
        # (parameters whose gradient is mathematically zero — k_proj biases: softmax shift invariance — carry rounding noise
        # of either sign, which AdamW turns into +-lr * g / eps ~ 1e-5 steps: hence the absolute floor of 5 % of one lr step)
        torch.testing.assert_close(res["plain_epi"][2], res["plain"][2], rtol=1e-5, atol=5e-5)
        # bf16 communication: gradients pass through one bf16 rounding (2^-9 relative per element)
        n_plain = res["plain"][1].item()
        assert abs(res["bf16"][1].item() - n_plain) <= 4e-3 * n_plain
        assert abs(res["bf16"][0][1] - res["plain"][0][1]) <= 2e-2 * abs(res["plain"][0][1])
    finally:
        if created:
            dist.destroy_process_group()


def test_memvla_through_trainer_norm_and_buckets(golden_dir):
    """retrieval / gate blocks run once per sample: their buckets must fold into the norm after the LAST sample's
    backward.  Tiny buckets so every block folds separately."""
    from tests.test_memvla_gpu import T, build
    g, cfg, m = build(golden_dir, "float32", True)
    m.train()
    tr = _trainer(m, min_bucket_bytes=1 << 10)
    fired = []
    hook = tr.norm_tracker.bucket_ready
    tr.norm_tracker.bucket_ready = lambda b: (fired.append(b), hook(b))
    batch = dict(input_ids=T(g["input_ids"]), attention_mask=T(g["attention_mask"]), images=T(g["images"]),
                 actions=T(g["actions"]), indexes=[list(map(int, r)) for r in g["indexes"]], noise=T(g["noise"]),
                 timesteps=T(g["timesteps"]), drop_ids=T(g["drop_u"]) < 0.1)
    loss = tr.step(batch)
    torch.cuda.synchronize()
    assert abs(loss.item() - float(g["loss"])) < 1e-3 * abs(float(g["loss"]))
    full = m.store.grad.double().norm().item()
    assert abs(tr.opt.norm.item() - full) <= 1e-6 * full, (tr.opt.norm.item(), full)
    assert abs(tr.opt.norm.item() - float(g["grad_norm"])) < 1e-3 * float(g["grad_norm"])
    assert len(fired) == len(set(fired)), "a bucket fired twice"


def test_bf16_gradient_arena_step_tracks_fp32_gradient_step(golden_dir):
    """grad_dtype=bfloat16 (the reference's DeepSpeed bf16 recipe: bf16 gradients, fp32 masters): the bf16 dW products write the
    bf16 gradient arena only, the other gradients are cast per slot, AdamW and the clip norm read that arena.  Against the
    fp32-gradient step of the same bf16-compute model: identical first loss, norm within one bf16 rounding per element,
    parameters after two steps within the AdamW step size, and the run is bit-reproducible"""
    g, cfg, w = load_golden(golden_dir, "t1")
    res = {}
    for tag, gd in (("f32", torch.float32), ("bf16", torch.bfloat16), ("bf16_again", torch.bfloat16)):
        m = build_product(cfg, w, "bfloat16", DEV, train=True)
        m.train()
        tr = _trainer(m, min_bucket_bytes=1 << 14, grad_dtype=gd)
        assert m.store.bf16_grads == (gd == torch.bfloat16)
        assert (tr.reducer is not None and tr.reducer.local_only) == (gd == torch.bfloat16)
        assert m.store.epi_sumsq
        losses = [tr.step(_batch(g)).item() for _ in range(2)]
        torch.cuda.synchronize()
        res[tag] = (losses, tr.opt.norm.item(), m.store.master.clone())
        if gd == torch.bfloat16:
            # the norm the clip used is the norm of the bf16 arena AdamW read
            full = m.store.gradc.double().norm().item()
            assert abs(tr.opt.norm.item() - full) <= 1e-5 * full, (tr.opt.norm.item(), full)
    assert res["bf16"][0][0] == res["f32"][0][0]
    assert abs(res["bf16"][1] - res["f32"][1]) <= 4e-3 * res["f32"][1]
    assert abs(res["bf16"][0][1] - res["f32"][0][1]) <= 2e-2 * abs(res["f32"][0][1])
    assert (res["bf16"][2] - res["f32"][2]).abs().max().item() <= 2.5e-3      # lr 1e-3: at most ~2 AdamW steps apart
    assert res["bf16_again"][0] == res["bf16"][0] and res["bf16_again"][1] == res["bf16"][1]
    assert torch.equal(res["bf16_again"][2], res["bf16"][2])


def test_overlapped_optimizer_is_bit_identical_to_the_serial_step(golden_dir):
    """NativeTrainer(overlap_optimizer=True): AdamW runs segment by segment in forward order on a side stream and the next
    step's forward waits per bucket (engine.FusedAdamW / ParamStore.wait_pending).  Same arithmetic, other schedule: losses,
    clip norm, masters, shadows and moments after three steps are bit-identical to the serial step — fp32 and bf16 compute,
    tiny segments so that every block is its own segment, followed by an inference request (the persistent DiT kernel reads
    the masters through raw pointers) and a state_dict() read."""
    g, cfg, w = load_golden(golden_dir, "t1")
    for dtype in ("float32", "bfloat16"):
        res = {}
        for tag, ov in (("serial", False), ("overlap", True)):
            m = build_product(cfg, w, dtype, DEV, train=True)
            m.train()
            from dexbotic_amd.engine import FusedAdamW
            tr = _trainer(m, min_bucket_bytes=1 << 14, overlap_optimizer=ov)
            if ov:
                tr.opt = FusedAdamW(m.store, tr.cfg, exclude=m.unused_parameter_names(), overlap=True, segment_elems=1)
                assert len(tr.opt.segments) > 8
            losses = [tr.step(_batch(g)).item() for _ in range(3)]
            if ov:
                assert m.store._pending, "the update of the last step should still be pending"
            m.eval()
            acts = m.inference_action(torch.from_numpy(g["infer_ids"]).to(DEV), torch.from_numpy(g["images"][:1]).to(DEV),
                                      {"cfg_scale": 1.5, "num_ddim_steps": 10,
                                       "action_norms": {"min": g["norm_min"].tolist(), "max": g["norm_max"].tolist()}},
                                      noise=torch.from_numpy(g["init_noise"]).to(DEV))
            tr.synchronize()
            sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
            torch.cuda.synchronize()
            res[tag] = (losses, tr.opt.norm.clone(), m.store.master.clone(), tr.opt.m.clone(), tr.opt.v.clone(),
                        None if m.store.shadow is None else m.store.shadow.clone(), np.asarray(acts), sd)
        a, b = res["serial"], res["overlap"]
        assert a[0] == b[0], (dtype, a[0], b[0])
        for i in (1, 2, 3, 4):
            assert torch.equal(a[i], b[i]), (dtype, i)
        if a[5] is not None:
            assert torch.equal(a[5], b[5])
        assert np.array_equal(a[6], b[6])
        assert all(torch.equal(a[7][k], b[7][k]) for k in a[7])


def test_native_avg_reduce_scatter_all_gather_in_place_at_world_one():
    """the call sequence every N > 1 RCCL run takes — reduce_scatter_tensor(op=AVG) into this rank's shard of the SAME
    buffer, all_gather_into_tensor back, 32-byte shard alignment, all-reduced tail — executed on hardware with one GPU
    (GradReducer(native_avg_world1=True); world size 1 otherwise swaps AVG for SUM).  The mean over one rank is the identity:
    bf16 and fp32 slices, lengths that do and do not divide, both algorithms, must come back bit-identical."""
    import torch.distributed as dist
    from dexbotic_amd.engine import GradReducer, ParamStore
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        for comm in (torch.bfloat16, torch.float32):
            for algo in ("rs_ag", "allreduce"):
                st = ParamStore(DEV, torch.float32)
                for i, n in enumerate((4096, 1000, 77, 12345)):
                    st.new_bucket()
                    st.register([(f"p{i}", (n,))])
                st.finalize(train=True)
                red = GradReducer(st, min_bucket_bytes=1 << 10, force=True, comm_dtype=comm, algo=algo, native_avg_world1=True)
                gen = torch.Generator(device=DEV).manual_seed(5)
                st.grad.copy_(torch.randn(st.total, device=DEV, generator=gen))
                want = st.grad.to(comm).clone()
                st.begin_step()
                for i in reversed(range(4)):
                    st.mark_written(f"p{i}")
                red.finish()
                torch.cuda.synchronize()
                got = red.result_arena
                for i in range(4):
                    sl = st.slots[f"p{i}"]
                    assert torch.equal(got[sl.offset:sl.offset + sl.numel], want[sl.offset:sl.offset + sl.numel]), (comm, algo, i)
                assert red.collectives >= 2 and red.bytes_reduced > 0
    finally:
        if created:
            dist.destroy_process_group()


def test_sharded_optimizer_step_at_world_one_over_rccl(golden_dir):
    """the sharded optimizer step (engine.ShardPlan, round 6) on RCCL with ONE rank: the plan of world size 1 owns every exchanged
    element, so reduce-scatter in place -> sum(g^2) over the 'shard' + scalar all-reduce -> adamw over the owned ranges with PACKED
    moments -> all-gather of the updated weights must reproduce the plain step (fp32 exchange; fp32 and bf16 compute): bit for bit
    with the clip out of the way (max_grad_norm 1e9 — the sharded step adds sum(g^2) slice by slice, the plain one over merged
    ranges: under the 1.0 clip the two norms differ in the last bits, 6.4568820 vs 6.4568825, and so does every parameter), within
    that rounding with it; and ``emulate_world=8`` (bench.py dp8_emulated_ms_per_step: ownership of rank 0 of 8, collectives at world size 1) moves exactly
    the owned eighth of the parameters — by exactly what the full step moves them."""
    import torch.distributed as dist
    g, cfg, w = load_golden(golden_dir, "t1")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29534")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        for dtype, clip in (("float32", 1e9), ("bfloat16", 1e9), ("float32", 1.0)):
            res = {}
            for tag, kw in (("plain", {}), ("sharded", dict(force_reducer=True, shard_optimizer=True)),
                            ("emulated8", dict(force_reducer=True, shard_optimizer=True, emulate_world=8))):
                m = build_product(cfg, w, dtype, DEV, train=True)
                m.train()
                tr = _trainer(m, clip=clip, min_bucket_bytes=1 << 14, **kw)
                m.store.epi_sumsq = False                     # the read-back sum of squares in every leg (one summation order)
                p0 = m.store.master.clone()
                losses = [tr.step(_batch(g)).item() for _ in range(2)]
                tr.consolidate()
                torch.cuda.synchronize()
                res[tag] = (losses, tr.opt.norm.clone(), m.store.master.clone(), None if m.store.shadow is None else m.store.shadow.clone(),
                            p0, tr)
            pl, sh, em = res["plain"], res["sharded"], res["emulated8"]
            tr_s, tr_e = sh[5], em[5]
            assert tr_s.sharded and tr_s.reducer.plan.world == 1 and tr_s.opt.chunk_mv_start is not None
            assert tr_s.reducer.bytes_gathered > 0 and tr_s.reducer.bytes_reduced > 0
            if clip == 1.0:
                assert abs(sh[1].item() - pl[1].item()) <= 1e-6 * pl[1].item() and sh[0][0] == pl[0][0]
                torch.testing.assert_close(sh[2], pl[2], rtol=1e-5, atol=5e-5)      # (floor: 5 % of one lr step, as above)
                continue
            assert sh[0] == pl[0], (dtype, sh[0], pl[0])
            assert abs(sh[1].item() - pl[1].item()) <= 1e-6 * pl[1].item(), dtype      # (the reported norm: summation order)
            dd = (sh[2] != pl[2])
            assert not bool(dd.any()), (f"sharded step at world 1 != plain step ({dtype}): {int(dd.sum())} elements, max abs "
                                        f"{float((sh[2] - pl[2]).abs().max()):.3e}, first at {int(torch.nonzero(dd)[0])}")
            if pl[3] is not None:
                assert torch.equal(sh[3], pl[3])
            # emulation: rank 0 of 8 — packed moments for an eighth of the exchanged elements (+ the replicated tails)
            plan = tr_e.reducer.plan
            assert plan.world == 8 and plan.rank == 0 and tr_e.reducer.world == 1
            owned = torch.zeros(p0.numel() if False else em[4].numel(), dtype=torch.bool, device=DEV)
            for a, b in plan.owned():
                owned[a:b] = True
            exch = sum(sl["hi"] - sl["lo"] for sl in plan.slices)
            assert int(owned.sum()) <= tr_e.opt.m.numel() <= int(owned.sum()) + 3 * len(plan.owned()) and int(owned.sum()) < 0.25 * exch
            # first step from identical state: the owned parameters move exactly as in the full step, the others not at all
            m1 = build_product(cfg, w, dtype, DEV, train=True)
            m1.train()
            t1 = _trainer(m1, clip=clip, min_bucket_bytes=1 << 14, force_reducer=True, shard_optimizer=True, emulate_world=8)
            m1.store.epi_sumsq = False
            m2 = build_product(cfg, w, dtype, DEV, train=True)
            m2.train()
            t2 = _trainer(m2, clip=clip, min_bucket_bytes=1 << 14)
            m2.store.epi_sumsq = False
            t1.step(_batch(g))
            t2.step(_batch(g))
            torch.cuda.synchronize()
            a1, a2, start = m1.store.master, m2.store.master, em[4]
            assert torch.equal(a1[owned], a2[owned])
            assert torch.equal(a1[~owned], start[~owned])
            assert not torch.equal(a2[~owned], start[~owned])
    finally:
        if created:
            dist.destroy_process_group()
