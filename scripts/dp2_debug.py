#!/usr/bin/env python
"""debug aid for tests/test_zz_dp2_gpu.py: the gradient arena after ONE backward (before the update) of the single-process run on
the concatenated batch against the exchanged arena of a 2-rank run, slot by slot"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import test_zz_dp2_gpu as T  # noqa: E402

NSTEP = int(os.environ.get("DP2_STEPS", "2"))     # the backward that is compared is the NSTEP-th


def worker(rank, world, port, tmp):
    import datetime
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    x = T._episodes()
    m, tr = T._build(distributed=True, grad_sync="allreduce", grad_reduce_op="sum")
    for _ in range(NSTEP - 1):
        tr.step(T._shard(x, list(range(rank, T.B, world))))
    loss = tr.micro_step(T._shard(x, list(range(rank, T.B, world))))
    torch.cuda.synchronize()
    g = tr.reducer.result_arena.float() * tr.reducer.grad_scale
    np.savez(os.path.join(tmp, f"dbg{rank}.npz"), loss=float(loss), grad=g.cpu().numpy(),
             sumsq=float(tr._sumsq.item()), arena_sumsq=float(tr.reducer.result_arena.double().pow(2).sum()), master=m.store.master.cpu().numpy())
    dist.destroy_process_group()


def main():
    import tempfile
    import torch.multiprocessing as mp
    tmp = tempfile.mkdtemp()
    x = T._episodes()
    m, tr = T._build()
    for _ in range(NSTEP - 1):
        tr.step(T._shard(x, list(range(T.B))))
    loss = tr.micro_step(T._shard(x, list(range(T.B))))
    torch.cuda.synchronize()
    ref = m.store.grad.float().cpu().numpy()
    ref_sumsq = float(tr._sumsq.item())
    ref_master = m.store.master.cpu().numpy()
    print("single: reported sumsq", ref_sumsq, "sumsq of the arena itself", float(m.store.grad.double().pow(2).sum()))
    slots = sorted(m.store.slots.values(), key=lambda s: s.offset)
    # per-episode losses, for the record
    print("full-batch loss", float(loss))
    del m, tr
    torch.cuda.empty_cache()
    mp.spawn(worker, args=(2, T._free_port(), tmp), nprocs=2, join=True)
    r = [np.load(os.path.join(tmp, f"dbg{k}.npz")) for k in range(2)]
    print("rank losses", float(r[0]["loss"]), float(r[1]["loss"]), "mean", (float(r[0]["loss"]) + float(r[1]["loss"])) / 2)
    print("sumsq: single", ref_sumsq, "ranks (of the SUM arena)", float(r[0]["sumsq"]), float(r[1]["sumsq"]), "-> mean-gradient sumsq", float(r[0]["sumsq"]) / 4)
    print("ranks identical:", np.array_equal(r[0]["grad"], r[1]["grad"]), " rank 0: sumsq of its arena", float(r[0]["arena_sumsq"]),
          " parameters before this backward: max |2-rank - single|", float(np.abs(r[0]["master"] - ref_master).max()))
    rows = []
    for s in slots:
        a, b = r[0]["grad"][s.offset:s.offset + s.numel], ref[s.offset:s.offset + s.numel]
        den = np.abs(b).max()
        rows.append((np.abs(a - b).max() / (den + 1e-30), den, s.name))
    rows.sort(reverse=True)
    for d, den, n in rows[:25]:
        print(f"{d:10.3e}  max|ref| {den:10.3e}  {n}")


if __name__ == "__main__":
    main()
