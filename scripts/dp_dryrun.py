#!/usr/bin/env python
"""Dry run of the multi-process launch contract of bench.py on CPU (no GPU here, the 8-GPU run is the driver's):

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P scripts/dp_dryrun.py

reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* exactly like bench.py, initialises the process group (gloo instead of
nccl = RCCL), shards synthetic "episodes" by rank (seed 1234 + rank, like bench.py), runs the GradReducer (RS+AG, fp32 and
bf16 exchange) over a ParamStore whose buckets fire in backward order, and checks that every rank ends with the mean
gradient and that the max-over-ranks timing all-reduce works.  Rank 0 prints one JSON line."""
import json
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    dist.init_process_group("gloo")
    assert dist.get_world_size() == world and dist.get_rank() == rank and 0 <= local_rank < world
    from dexbotic_amd.engine import GradReducer, ParamStore
    st = ParamStore("cpu", torch.float32)
    names = []
    for b in range(6):
        st.new_bucket()
        grp = [(f"blk{b}.w", (33, 17)), (f"blk{b}.b", (17,))]
        st.register(grp)
        names += [n for n, _ in grp]
    st.finalize(train=True)
    st.set_expected(())
    ok = True
    t0 = time.perf_counter()
    for comm in (torch.float32, torch.bfloat16):
        red = GradReducer(st, min_bucket_bytes=1024, comm_dtype=comm, algo="rs_ag")
        torch.manual_seed(1234 + rank)                       # rank-sharded synthetic episodes
        st.begin_step()
        st.on_bucket_ready = red.bucket_ready
        local = {}
        for n in reversed(names):
            g = torch.randn(st.slots[n].shape)
            st.g(n).copy_(g)
            local[n] = g
            st.mark_written(n)
        red.finish()
        gathered = [None] * world
        dist.all_gather_object(gathered, local)
        arena = red.result_arena
        for n in names:
            s = st.slots[n]
            got = arena[s.offset:s.offset + s.numel].float().view(s.shape) * red.grad_scale   # SUM exchange: 1 / world rides in the clip coefficient
            mean = sum((g[n].to(comm).float() if comm == torch.bfloat16 else g[n]) for g in gathered) / world
            # bf16 exchange: every partial sum of the ring is rounded to bf16 (2^-9 of the running sum per step)
            ok_n = bool(torch.allclose(got, mean, rtol=2e-2, atol=4e-2)) if comm == torch.bfloat16 else \
                bool(torch.allclose(got, mean, rtol=1e-5, atol=1e-6))
            if not ok_n:
                print(f"rank {rank}: {n} {comm} max abs err {(got - mean).abs().max().item():.3e}", flush=True)
            ok &= ok_n
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.barrier()
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)               # bench.py reports the max over ranks
    flag = torch.tensor([1.0 if ok else 0.0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        print(json.dumps({"dryrun": "ok" if flag.item() == 1.0 else "MISMATCH", "world": world, "max_seconds": round(dt.item(), 3)}),
              flush=True)
    dist.destroy_process_group()
    sys.exit(0 if flag.item() == 1.0 else 1)


if __name__ == "__main__":
    main()
