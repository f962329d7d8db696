#!/usr/bin/env python
"""debug: 2-rank sharded step on one GPU (gloo) — where do a rank's masters differ before / after gather_masters()?"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, port):
    import datetime
    import torch.distributed as dist
    from tests import test_zz_dp2_gpu as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    x = D._episodes()
    dtype = os.environ.get("DT", "float32")
    m, tr = D._build(grad_accum=1, dtype=dtype, clip=1.0, distributed=True, grad_sync="rs_ag", grad_reduce_op="sum", shard_optimizer=True)
    st, red = m.store, tr.reducer
    plan = red.plan
    mine = list(range(rank, D.B, world))
    orig_gp = red.gather_params

    def gp(overlap=True):
        torch.cuda.synchronize()
        own = plan.owned()
        print(f"[rank {rank}] before gather_params: nan master {int(torch.isnan(st.master).sum())} (own {sum(int(torch.isnan(st.master[a:b]).sum()) for a, b in own)}) "
              f"nan v {int(torch.isnan(tr.opt.v).sum())} min v {float(tr.opt.v.min()):.3e} nan m {int(torch.isnan(tr.opt.m).sum())} sumsq {float(tr._sumsq):.6e} "
              f"chunks {tr.opt.chunk_start.numel()} m numel {tr.opt.m.numel()} max mv end {int((tr.opt.chunk_mv_start + tr.opt.chunk_len).max())}", flush=True)
        orig_gp(overlap)
        torch.cuda.synchronize()
        print(f"[rank {rank}] after gather_params: nan master {int(torch.isnan(st.master).sum())}", flush=True)
    red.gather_params = gp
    for step in range(3):
        loss = tr.step(D._shard(x, mine))
        torch.cuda.synchronize()
        before = st.master.clone()
        print(f"[rank {rank}] step {step}: loss {float(loss):.6f} norm {float(tr.opt.norm):.6f} coef {float(tr.opt.coef):.6f} nan master {int(torch.isnan(before).sum())} "
              f"nan m {int(torch.isnan(tr.opt.m).sum())} nan grad(own) {sum(int(torch.isnan(st.grad[a:b]).sum()) for a, b in plan.owned())}", flush=True)
        red.gather_masters()
        torch.cuda.synchronize()
        after = st.master
        diff = (before != after)
        if rank == 1 or True:
            print(f"[rank {rank}] step {step}: {int(diff.sum())} master elements changed by gather_masters; w32 buckets {sorted(st._w32_buckets)}", flush=True)
            for i, sl in enumerate(plan.slices):
                d = diff[sl["lo"]:sl["hi"]]
                if int(d.sum()):
                    own = plan.shard(i)
                    idx = torch.nonzero(d).flatten() + sl["lo"]
                    j = int(idx[0])
                    print(f"      first at {j} (slice lo {sl['lo']}, own {own}): before {float(before[j])!r} after {float(after[j])!r}", flush=True)
                    in_own = int(((idx >= own[0]) & (idx < own[1])).sum())
                    in_tail = int((idx >= sl["lo"] + sl["body"]).sum())
                    names = sorted({s.name for s in st.slots.values() if s.bucket in sl["buckets"]})[:3]
                    print(f"   slice {i} buckets {sl['buckets']} per {sl['per']}: {int(d.sum())} differ ({in_own} in own shard, {in_tail} in tail) e.g. {names}", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    from tests.test_zz_dp2_gpu import _free_port
    mp.spawn(worker, args=(2, _free_port()), nprocs=2, join=True)
