#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/pmc_w4
cd /tmp && export TMPDIR=/tmp; cd $R
O=gpurun_out/pmc_w4
timeout 300 rocprofv3 --kernel-trace -d $O -o k -- python scripts/w4_check.py pmc > $O/k.log 2>&1
python - <<'PY'
import sqlite3
db = sqlite3.connect("gpurun_out/pmc_w4/k_results.db"); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
print([t for t in tabs if 'kernel' in t.lower()][:20])
for t in tabs:
    if 'kernel' in t.lower():
        cols = [r[1] for r in cur.execute(f"pragma table_info({t})")]
        print(t, cols)
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
sel = [c for c in cols if any(k in c.lower() for k in ("name", "lds", "vgpr", "sgpr", "scratch", "grid", "workgroup", "accum"))]
for row in cur.execute(f"select distinct {','.join(sel)} from kernels where name like '%Cijk%' or name like '%gemm_%' limit 12"):
    print(dict(zip(sel, [str(x)[:70] for x in row])))
PY
rm -f $O/*.db
