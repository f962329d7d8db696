#!/usr/bin/env python
"""debug: plain step vs sharded step at world 1 (RCCL, forced) — first quantity that differs"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch.distributed as dist  # noqa: E402
from tests.helpers import build_product, load_golden  # noqa: E402
from tests.test_parity_gpu import _batch  # noqa: E402
from tests.test_zz_dp_gpu import _trainer  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
g, cfg, w = load_golden(os.path.join(ROOT, "tests", "golden"), "t1")
runs = {}
for tag, kw in (("plain", {}), ("sharded", dict(force_reducer=True, shard_optimizer=True)), ("repl", dict(force_reducer=True, shard_optimizer=False))):
    m = build_product(cfg, w, "float32", "cuda", train=True)
    m.train()
    tr = _trainer(m, min_bucket_bytes=1 << 14, **kw)
    m.store.epi_sumsq = False
    snaps = []
    for step in range(2):
        tr.step(_batch(g))
        torch.cuda.synchronize()
        fm, fv = tr.opt.full_moments()
        snaps.append(dict(master=m.store.master.clone(), grad=m.store.grad.clone(), m=fm.clone(), v=fv.clone(), norm=float(tr.opt.norm), coef=float(tr.opt.coef)))
    runs[tag] = (snaps, m, tr)
st = runs["plain"][1].store
names = sorted(st.slots.values(), key=lambda s: s.offset)
for other in ("sharded", "repl"):
    for step in range(2):
        a, b = runs["plain"][0][step], runs[other][0][step]
        print(f"== plain vs {other}, after step {step}: norm {a['norm']!r} vs {b['norm']!r}  coef {a['coef']!r} vs {b['coef']!r}")
        for key in ("grad", "m", "v", "master"):
            d = a[key] != b[key]
            n = int(d.sum())
            if n:
                j = int(torch.nonzero(d)[0])
                slot = next(s for s in reversed(names) if s.offset <= j)
                print(f"   {key}: {n} differ, max abs {float((a[key] - b[key]).abs().max()):.3e}, first at {j} ({slot.name} + {j - slot.offset}): {float(a[key][j])!r} vs {float(b[key][j])!r}")
            else:
                print(f"   {key}: equal")
pl = runs["sharded"][2].reducer.plan
print("slices", [(sl["lo"], sl["hi"], sl["per"], sl["buckets"]) for sl in pl.slices][:6], "owned", pl.owned()[:6])
o = runs["sharded"][2].opt
print("chunks", o.chunk_start[:8].tolist(), o.chunk_len[:8].tolist(), o.chunk_mv_start[:8].tolist(), "ranges", o.ranges[:4], o._range_base[:4])
dist.destroy_process_group()
