#!/bin/bash
# L2-miss bytes of the decoder's prefill products at 287 rows (FETCH_SIZE x 2, MI355X_MICROARCH.md): how much of the few-row
# kernels' time is fabric-side traffic?
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc4; mkdir -p $O
ROWS=287 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O -o fetch -- python scripts/prefill_gemm_bench.py > $O/fetch.log 2>&1
python profiles/rocpd_stats.py --pmc $O/fetch_results.db "gemm_nt_t128,gemm_nt_ring" > $O/fetch.txt 2>&1
rm -f $O/*.db
cat $O/fetch.txt | cut -c1-200 | head -12
grep "M=" $O/fetch.log
