#!/bin/bash
# pi0 per-request inference kernel table (difference of traces with 6 and 16 requests)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/prof
cd /tmp; export TMPDIR=/tmp
for n in 6 16; do INFER_ONLY=1 REQS=$n rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o pi$n -- python $R/scripts/pi0_bench.py > $R/gpurun_out/r03_pi0_infer_$n.log 2>&1; done
cd $R
python profiles/rocpd_stats.py --per-step gpurun_out/prof/pi6_results.db 6 gpurun_out/prof/pi16_results.db 16 > gpurun_out/r03_pi0_infer_kernel_stats.txt
head -40 gpurun_out/r03_pi0_infer_kernel_stats.txt | cut -c1-170
grep "^{" gpurun_out/r03_pi0_infer_16.log | tail -1
rm -rf gpurun_out/prof
