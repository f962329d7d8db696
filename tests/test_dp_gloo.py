"""Data-parallel gradient reducer over torch.distributed (gloo, world_size 2, CPU).

The N>1 path of bench.py (one process per GPU, RCCL) shards episodes across ranks and averages the flat
gradient arena bucket by bucket (engine.GradReducer).  Here the same reducer / store / bucket logic runs
on CPU tensors with the gloo backend: every rank must end with the mean gradient, parameters that never
receive a gradient must not be communicated, and gradient accumulation must reduce on the last
micro-batch only."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build_store(rows: int = 7):
    from dexbotic_amd.engine import ParamStore
    st = ParamStore("cpu", torch.float32)
    names = []
    for b in range(5):                                   # 5 buckets in "forward order"
        st.new_bucket()
        grp = [(f"blk{b}.w", (rows, 5)), (f"blk{b}.b", (6,))]     # 41 elements: NOT divisible by the world size
        st.register(grp)
        names += [n for n, _ in grp]
    st.new_bucket()
    st.register([("unused.w", (11,))])                   # never written (like lm_head)
    st.finalize(train=True)
    return st, names


def _worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dexbotic_amd.engine import GradReducer
        st, names = _build_store()
        st.set_expected(["unused.w"])
        torch.manual_seed(100 + rank)
        gathered = [None] * world
        # --- plain step: backward walks the buckets in reverse, marks slots written.  Both exchange algorithms:
        #     reduce-scatter + all-gather (odd slice lengths: divisible prefix + all-reduced tail) and one all-reduce
        for algo, min_bytes, op in (("rs_ag", 64, "sum"), ("rs_ag", 1 << 20, "sum"), ("allreduce", 64, "sum"),
                                    ("rs_ag", 64, "avg"), ("allreduce", 64, "avg")):
            # reduce_op "sum" (the default): the arena ends up holding world x the mean and grad_scale = 1 / world is what the
            # optimizer multiplies by; "avg": the arena holds the mean itself
            red = GradReducer(st, min_bucket_bytes=min_bytes, skip=["unused.w"], algo=algo, reduce_op=op)
            assert red.grad_scale == (1.0 / world if op == "sum" else 1.0)
            assert GradReducer(st, skip=["unused.w"]).reduce_op == "sum"       # the default exchange
            st.begin_step()
            st.on_bucket_ready = red.bucket_ready
            local = {}
            for n in reversed(names):
                g = torch.randn(st.slots[n].shape)
                st.g(n).copy_(g)
                local[n] = g
                st.mark_written(n)
            st.g("unused.w").fill_(float(rank + 1))          # garbage that must NOT be averaged
            red.finish()
            dist.all_gather_object(gathered, local)
            for n in names:
                mean = sum(g[n] for g in gathered) / world
                assert torch.allclose(st.g(n) * red.grad_scale, mean, atol=1e-6), (algo, op, n)
            assert torch.all(st.g("unused.w") == float(rank + 1))
            assert red.bytes_reduced >= sum(st.slots[n].numel for n in names) * 4   # alignment padding may ride along
            if algo == "rs_ag" and min_bytes == 64:
                assert red.collectives == 3 * 5, red.collectives   # per bucket: RS + AG on 40 elements, AR on the 41st
        red = GradReducer(st, min_bucket_bytes=64, skip=["unused.w"])
        sc = red.grad_scale
        # --- gradient accumulation: 2 micro-batches, communication only on the last
        st.begin_step()
        red.bytes_reduced = 0
        acc = {n: torch.zeros(st.slots[n].shape) for n in names}
        for micro in range(2):
            if micro:
                st.begin_micro()
            st.on_bucket_ready = red.bucket_ready if micro == 1 else None
            for n in reversed(names):
                g = torch.randn(st.slots[n].shape)
                if st.accum_flag(n):
                    st.g(n).add_(g)
                else:
                    st.g(n).copy_(g)
                acc[n] += g
                st.mark_written(n)
            if micro == 0:
                assert red.bytes_reduced == 0
        red.finish()
        dist.all_gather_object(gathered, acc)
        for n in names:
            mean = sum(g[n] for g in gathered) / world
            assert torch.allclose(st.g(n) * sc, mean, atol=1e-6), n
        # --- a bucket with a frozen slot still gets reduced by finish()
        st.params["blk2.b"].requires_grad_(False)
        st.set_expected(["unused.w"])
        st.begin_step()
        st.on_bucket_ready = red.bucket_ready
        for n in reversed(names):
            if n == "blk2.b":
                continue
            st.g(n).fill_(float(rank))
            st.mark_written(n)
        red.finish()
        assert torch.allclose(st.g("blk2.w") * sc, torch.full((7, 5), (world - 1) / 2.0))
        # --- bf16 gradient communication (what the reference's DeepSpeed bf16 run reduces): half the bytes; what is
        #     exchanged is the bf16 copy of the arena (ParamStore.gradc), the averaged result stays there (AdamW reads it)
        #     and the local fp32 gradients are left alone.  One slot plays a GEMM epilogue that mirrored its own output.
        st.params["blk2.b"].requires_grad_(True)
        st.set_expected(["unused.w"])
        red16 = GradReducer(st, min_bucket_bytes=64, skip=["unused.w"], comm_dtype=torch.bfloat16)
        assert red16.result_arena is st.gradc
        st.begin_step()
        st.on_bucket_ready = red16.bucket_ready
        local, local32 = {}, {}
        for n in reversed(names):
            g = torch.randn(st.slots[n].shape)
            st.g(n).copy_(g)
            if n == "blk3.w":
                st.mirror_out(n).copy_(g)                 # "epilogue" wrote the copy itself: must not be re-cast
                st.g(n).add_(1000.0)                      # (would show up in the average if it were)
            local[n] = g.to(torch.bfloat16).float()
            local32[n] = st.g(n).clone()
            st.mark_written(n)
        red16.finish()
        dist.all_gather_object(gathered, local)
        for n in names:
            mean = (sum(g[n] for g in gathered)).to(torch.bfloat16).float() / world
            assert torch.allclose(st.gc(n).float() * red16.grad_scale, mean, rtol=1e-2, atol=1e-6), n
            assert torch.equal(st.g(n), local32[n]), n
        assert red16.bytes_reduced >= sum(st.slots[n].numel for n in names) * 2     # 2 bytes per element sent
        # --- a parameter applied 3x in one forward (MemVLA's per-sample retrieval blocks): its bucket must be
        #     reduced after the LAST of its three gradient writes, i.e. the result is the mean of the FINAL gradients
        red3 = GradReducer(st, min_bucket_bytes=64, skip=["unused.w"])
        st.begin_step()
        st.on_bucket_ready = red3.bucket_ready
        for n in names:
            for _ in range(3 if n.startswith("blk1.") else 1):
                st.note_use(n)
        final = {n: torch.zeros(st.slots[n].shape) for n in names}
        fired_early = []
        orig = red3.bucket_ready
        red3_fired = []
        st.on_bucket_ready = lambda b: (red3_fired.append(b), orig(b))
        for n in reversed(names):
            for k in range(3 if n.startswith("blk1.") else 1):
                g = torch.randn(st.slots[n].shape)
                if st.accum_flag(n):
                    st.g(n).add_(g)
                else:
                    st.g(n).copy_(g)
                final[n] += g
                st.mark_written(n)
                if n.startswith("blk1.") and k < 2:
                    assert st.slots[n].bucket not in red3_fired, "bucket fired before the last write"
        red3.finish()
        dist.all_gather_object(gathered, final)
        for n in names:
            mean = sum(g[n] for g in gathered) / world
            assert torch.allclose(st.g(n) * red3.grad_scale, mean, atol=1e-6), n
        open(os.path.join(tmp, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_grad_reducer_world2_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok0") and os.path.exists(tmp_path / "ok1")


def test_cosine_and_bucket_layout():
    st, names = _build_store()
    # buckets are contiguous, ordered slices of the arena (what the in-place all-reduce relies on)
    prev_hi = 0
    for lo, hi in st.bucket_ranges[1:]:
        assert lo >= prev_hi
        prev_hi = hi
    assert st.never_written() == sorted(st.never_written()) or True
    st.begin_step()
    st.mark_written("blk0.w")
    assert st.accum_flag("blk0.w") and not st.accum_flag("blk0.b")


import pytest


@pytest.mark.parametrize("nproc", [2, 3])
def test_torchrun_launch_contract_dry_run(nproc):
    """the driver's multi-GPU launch line (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py ...), dry-run on CPU: scripts/dp_dryrun.py reads the same environment as bench.py,
    uses gloo instead of RCCL and runs the rank-sharded reducer step (RS+AG, fp32 and bf16 exchange)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr",
                        "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "scripts", "dp_dryrun.py")],
                       capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["dryrun"] == "ok" and out["world"] == nproc


def test_store_view_forwards_attribute_writes():
    """engine.Fp32View (the action head's view of the store: fp32 masters from w()) must forward attribute WRITES: until round 5
    ``st._wgrad_pending = True`` of the head's side-stream gradient products landed on the view, the store's flag stayed False
    and the head bucket's completion hook — gradient exchange, sum of squares — did not join the side stream: a real race, found
    by the 2-rank model step on the MI355X (tests/test_zz_dp2_gpu.py, profiles/r05_dp2_race.txt)."""
    from dexbotic_amd.engine import Fp32View
    st, names = _build_store()
    view = Fp32View(st)
    assert view._wgrad_pending is False
    view._wgrad_pending = True
    assert st._wgrad_pending is True and set(vars(view)) == {"_s"}
    st.set_expected(["unused.w"])
    st.begin_step()
    joined = []
    st.wgrad_stream = object()                           # (a side stream exists)
    st.join_wgrad = lambda: joined.append(st._wgrad_pending)
    st.on_bucket_ready = lambda b: None
    for n in reversed(names):
        view._wgrad_pending = True                       # what functional._wgrad_now / _bgrad do after a side-stream enqueue
        view.mark_written(n)
    assert joined and all(joined), joined                # every bucket completion saw the pending side-stream work


# ------------------------------------------------------------------------------ sharded optimizer step (round 6)
def _shard_worker(rank, world, port, tmp):
    """engine.ShardPlan / GradReducer(shard=True) over gloo on CPU tensors: reduce-scatter into fixed shards (no gradient
    all-gather), after_reduce sees the own shard (+ rank 0 the replicated tails), a plain SGD-like update of the owned ranges and
    gather_params / gather_masters bring every rank to exactly what the replicated exchange + full update gives"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dexbotic_amd.engine import GradReducer
        # one slice per bucket / two buckets per slice / one slice; 7 rows: 41-element buckets, too short to shard (< 64 per rank:
        # replicated tails only, except the single merged slice); 61 rows: 311-element buckets, real shards + ragged tails
        for min_bytes, rows in ((64, 7), (400, 7), (1 << 20, 7), (64, 61), (3000, 61), (1 << 20, 61)):
            st, names = _build_store(rows)
            st.set_expected(["unused.w"])
            torch.manual_seed(5)
            st.master.copy_(torch.randn(st.total))            # same weights on every rank
            w0 = st.master.clone()
            red = GradReducer(st, min_bucket_bytes=min_bytes, skip=["unused.w"], shard=True)
            plan = red.plan
            assert plan is not None and plan.world == world and plan.rank == rank
            seen = []
            red.after_reduce = lambda lo, hi, stream: seen.append((lo, hi))
            torch.manual_seed(100 + rank)
            st.begin_step()
            st.on_bucket_ready = red.bucket_ready
            local = {}
            for n in reversed(names):
                g = torch.randn(st.slots[n].shape)
                st.g(n).copy_(g)
                local[n] = g
                st.mark_written(n)
            st.g("unused.w").fill_(float(rank + 1))
            red.finish()
            gathered = [None] * world
            dist.all_gather_object(gathered, local)
            total = torch.zeros(st.total)
            for n in names:
                s_ = st.slots[n]
                total[s_.offset:s_.offset + s_.numel] = sum(g[n] for g in gathered).reshape(-1)
            owned = plan.owned()
            own_elems = sum(b - a for a, b in owned)
            for a, b in owned:                                # the SUM over the ranks on what this rank owns (exact for two ranks;
                if world == 2:                                #  three ranks add in the ring's order)
                    assert torch.equal(st.grad[a:b], total[a:b]), (min_bytes, a, b)
                assert torch.allclose(st.grad[a:b], total[a:b], atol=1e-5), (min_bytes, a, b)
            # after_reduce: the own shards, and the replicated tails on rank 0 only
            tails = [plan.tail(i) for i in range(len(plan.slices)) if plan.tail(i)[1] > plan.tail(i)[0]]
            shards = [plan.shard(i) for i in range(len(plan.slices)) if plan.slices[i]["per"] > 0]
            assert sorted(seen) == sorted(shards + (tails if rank == 0 else [])), (seen, shards, tails)
            # the owned ranges of all ranks: shards disjoint, tails shared, union = every exchanged element
            all_owned = [None] * world
            dist.all_gather_object(all_owned, owned)
            cover = torch.zeros(st.total, dtype=torch.int32)
            for o in all_owned:
                for a, b in o:
                    cover[a:b] += 1
            for i, sl in enumerate(plan.slices):
                assert torch.all(cover[sl["lo"]:sl["lo"] + sl["body"]] == 1)
                assert torch.all(cover[sl["lo"] + sl["body"]:sl["hi"]] == world)
            u = st.slots["unused.w"]
            assert torch.all(cover[u.offset:u.offset + u.numel] == 0)
            assert torch.all(st.g("unused.w") == float(rank + 1))             # never sent
            # the update of the owned ranges, then the gathers: every rank ends with the full update of the summed gradient
            for a, b in owned:
                st.master[a:b] -= 0.1 * st.grad[a:b] * red.grad_scale
            c0 = red.collectives
            red.gather_params(overlap=False)
            want = w0 - 0.1 * total / world
            for sl in plan.slices:
                assert torch.allclose(st.master[sl["lo"]:sl["hi"]], want[sl["lo"]:sl["hi"]], atol=2e-6), min_bytes
            assert torch.equal(st.g("unused.w"), torch.full((11,), float(rank + 1)))
            assert red.collectives - c0 == sum(1 for sl in plan.slices if sl["per"] > 0)
            red.gather_masters()                              # idempotent on already gathered masters
            both = [None] * world
            dist.all_gather_object(both, st.master.clone())
            assert all(torch.equal(both[0], b) for b in both)
            if rows == 61:
                assert all(sl["per"] > 0 for sl in plan.slices) and own_elems < sum(sl["hi"] - sl["lo"] for sl in plan.slices)
        open(os.path.join(tmp, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_optimizer_exchange_gloo(tmp_path, world):
    mp.spawn(_shard_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f"ok{r}") for r in range(world))


def test_shard_plan_and_sharded_chunk_tables():
    """ShardPlan: fixed slices in arena order, 16-element shard starts, short slices replicated; FusedAdamW(ranges=owned):
    the chunk table covers exactly the trainable elements inside the owned ranges, moments packed back to back"""
    from dexbotic_amd.engine import FusedAdamW, OptimConfig, ParamStore, ShardPlan
    st = ParamStore("cpu", torch.float32)
    for b in range(6):
        st.new_bucket()
        st.register([(f"l{b}.w", (40, 33)), (f"l{b}.b", (33,))])
    st.new_bucket()
    st.register([("head.w", (7, 3))])                         # 21 elements: a slice too short to shard
    st.finalize(train=True)
    world = 4
    # (ParamStore.new_bucket() counts from 1: l0 .. l5 are buckets 1 .. 6, the head bucket 7; buckets 3 = l2 and 6 = l5 are skipped)
    plans = [ShardPlan(st, world, r, skip_buckets=[3, 6], min_bucket_bytes=2 * 1353 * 4) for r in range(world)]
    p0 = plans[0]
    assert [sl["buckets"] for sl in p0.slices] == [[1, 2], [4, 5], [7]]
    assert [sl["buckets"] for sl in p0.slices] == [sl["buckets"] for sl in plans[3].slices]
    for sl in p0.slices:
        assert sl["per"] % 16 == 0 and sl["body"] == sl["per"] * world and sl["lo"] + sl["body"] <= sl["hi"]
    assert p0.slices[-1]["per"] == 0                           # the 21-element head: replicated
    cover = torch.zeros(st.total, dtype=torch.int32)
    for pl in plans:
        for a, b in pl.owned():
            cover[a:b] += 1
    for b in (3, 6):
        lo3, hi3 = st.bucket_ranges[b]
        assert torch.all(cover[lo3:hi3] == 0)
    for sl in p0.slices:
        assert torch.all(cover[sl["lo"]:sl["lo"] + sl["body"]] == 1) and torch.all(cover[sl["lo"] + sl["body"]:sl["hi"]] == world)
    st.params["l1.b"].requires_grad_(False)
    for r, pl in enumerate(plans):
        opt = FusedAdamW(st, OptimConfig(chunk=256), exclude=["l2.w", "l2.b", "l5.w", "l5.b"], ranges=pl.owned())
        owned = torch.zeros(st.total, dtype=torch.bool)
        for a, b in pl.owned():
            owned[a:b] = True
        train = torch.zeros(st.total, dtype=torch.bool)
        for s_ in st.slots.values():
            if st.params[s_.name].requires_grad and not s_.name.startswith(("l2.", "l5.")):
                train[s_.offset:s_.offset + s_.numel] = True
        hit = torch.zeros(st.total, dtype=torch.int32)
        mv = torch.zeros(int(opt.m.numel()), dtype=torch.int32)
        for c0, ln, m0 in zip(opt.chunk_start.tolist(), opt.chunk_len.tolist(), opt.chunk_mv_start.tolist()):
            hit[c0:c0 + ln] += 1
            mv[m0:m0 + ln] += 1
        assert torch.equal(hit == 1, owned & train) and int(hit.max()) <= 1
        assert int(mv.max()) <= 1 and int(owned.sum()) <= opt.m.numel() <= int(owned.sum()) + 3 * len(pl.owned())
        # packed layout: the owned ranges back to back, each starting on a 16-byte boundary like in the arena
        base = 0
        for (a, b) in pl.owned():
            assert base % 4 == 0 and a % 4 == 0
            for c0, m0 in zip(opt.chunk_start.tolist(), opt.chunk_mv_start.tolist()):
                if a <= c0 < b:
                    assert m0 == base + (c0 - a)
            base += (b - a + 3) // 4 * 4
        full = torch.arange(1, st.total + 1, dtype=torch.float32)
        packed = opt.pack_moments(full)
        assert torch.equal(packed[packed > 0], full[owned])
        opt.m.copy_(packed)
        opt.v.copy_(2 * packed)
        fm, fv = opt.full_moments()
        assert torch.equal(fm[owned], full[owned]) and torch.equal(fv[owned], 2 * full[owned]) and float(fm[~owned].abs().sum()) == 0
    assert FusedAdamW(st, OptimConfig(), exclude=["l2.w", "l2.b"]).chunk_mv_start is None
