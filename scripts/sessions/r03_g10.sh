#!/bin/bash
# MemVLA per-frame inference kernel table (difference of traces with 8 and 20 frames), final kernels
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/prof
cd /tmp; export TMPDIR=/tmp
for n in 8 20; do SKIP_TRAIN=1 FRAMES=$n rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o mv$n -- python $R/scripts/memvla_bench.py > $R/gpurun_out/r03_memvla_infer_$n.log 2>&1; done
cd $R
python profiles/rocpd_stats.py --per-step gpurun_out/prof/mv8_results.db 8 gpurun_out/prof/mv20_results.db 20 > gpurun_out/r03_memvla_infer_kernel_stats.txt
head -14 gpurun_out/r03_memvla_infer_kernel_stats.txt | cut -c1-170
grep "^{" gpurun_out/r03_memvla_infer_20.log | tail -1
rm -rf gpurun_out/prof
