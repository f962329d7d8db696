#!/bin/bash
set -x
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/pmc_w4
timeout 600 python scripts/w4_check.py check time lib > gpurun_out/r04_w4_check2.log 2>&1; echo "rc $?" >> gpurun_out/r04_w4_check2.log
grep -v "^ok" gpurun_out/r04_w4_check2.log
cd /tmp && export TMPDIR=/tmp; cd $R
O=gpurun_out/pmc_w4
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $O -o sq -- python scripts/w4_check.py pmc > $O/sq.log 2>&1
python profiles/rocpd_stats.py --pmc $O/sq_results.db "gemm_pp_kernel,gemm_w4_kernel,Cijk" > gpurun_out/r04_pmc_w4_sq.txt 2>&1
python - <<'PY'
import sqlite3
cur = sqlite3.connect("gpurun_out/pmc_w4/sq_results.db").cursor()
for n, c, a in cur.execute("select name, count(*), avg(end-start) from kernels group by name order by sum(end-start) desc").fetchall()[:8]:
    print(f"{c:4d} {a/1e3:9.1f} us  {n[:120]}")
PY
rm -f $O/*.db
cat gpurun_out/r04_pmc_w4_sq.txt | cut -c1-200 | head -70
