#!/usr/bin/env python
"""Probe behind profiles/r05_dp2_race.txt: two ranks on one MI355X (gloo), per optimizer step the tracker's folded sum of squares against the
sum of squares of the exchanged arena itself.  DP2_SYNC=0: as the trainer runs it (no host synchronise between backward and update);
DP2_N: steps; DP2_SLOTS=1: per-slot sums of the action head's bucket at the end."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import test_zz_dp2_gpu as T  # noqa: E402


def worker(rank, world, port, tmp):
    import datetime
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    x = T._episodes()
    m, tr = T._build(distributed=True, grad_sync="allreduce", grad_reduce_op="sum")
    st, red = m.store, tr.reducer
    folds = []
    orig = tr.norm_tracker.fold

    def fold(lo, hi, after=None):
        folds.append((lo, hi))
        return orig(lo, hi, after)
    tr.norm_tracker.fold = fold
    red.after_reduce = lambda lo, hi, stream: tr.norm_tracker.fold(lo, hi, stream)
    out = []
    SYNC = os.environ.get("DP2_SYNC", "1") == "1"
    for s in range(int(os.environ.get('DP2_N', '3'))):
        folds.clear()
        tr.micro_step(T._shard(x, list(range(rank, T.B, world))))
        if not SYNC:
            tr.apply_update()
            torch.cuda.synchronize()
            arena = st.grad.double()
            out.append((float(tr._sumsq.item()), float(arena.pow(2).sum()), sum(float(arena[lo:hi].pow(2).sum()) for lo, hi in folds), [],
                        float(tr.opt.norm.item()), [round(float(arena[lo:hi].pow(2).sum()), 4) for lo, hi in folds]))
            continue
        torch.cuda.synchronize()
        rep = float(tr._sumsq.item())
        arena = st.grad.double()
        whole = float(arena.pow(2).sum())
        per = [float(arena[lo:hi].pow(2).sum()) for lo, hi in folds]
        tr.apply_update()
        torch.cuda.synchronize()
        out.append((rep, whole, sum(per), list(folds), float(tr.opt.norm.item()), per))
    if rank == 0 and os.environ.get("DP2_SLOTS"):
        arena = st.grad.double()
        for sl in sorted(st.slots.values(), key=lambda q: q.offset):
            if sl.offset >= 1887360:
                print(f"slot {sl.name:60s} {float(arena[sl.offset:sl.offset + sl.numel].pow(2).sum()):.6e}")
    if rank == 0:
        for s, (rep, whole, sp, fl, nm, per) in enumerate(out):
            print(f"step {s + 1}: tracker {rep:.6f}  whole arena {whole:.6f}  sum over the folded slices {sp:.6f}  norm {nm:.6f}  per slice {[round(v, 4) for v in per]}")
        print("bucket ranges", st.bucket_ranges, "skip", sorted(red.skip_buckets), "total", st.total)
    dist.destroy_process_group()


if __name__ == "__main__":
    import tempfile
    import torch.multiprocessing as mp
    mp.spawn(worker, args=(2, T._free_port(), tempfile.mkdtemp()), nprocs=2, join=True)
