#!/bin/bash
# MemVLA per-frame inference: cached perceptual K/V vs per-step, and the per-frame kernel table; pi0 per-step kernel table
set -x
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/prof
timeout 600 python -m pytest tests/test_memvla_gpu.py -x -q 2>&1 | tail -3
SKIP_TRAIN=1 NO_KV_CACHE=1 python scripts/memvla_bench.py 2>&1 | tail -1
SKIP_TRAIN=1 python scripts/memvla_bench.py 2>&1 | tail -1
cd /tmp; export TMPDIR=/tmp
for n in 8 20; do SKIP_TRAIN=1 FRAMES=$n rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o mv$n -- python $R/scripts/memvla_bench.py > $R/gpurun_out/r03_memvla_infer_$n.log 2>&1; done
cd $R
python profiles/rocpd_stats.py --per-step gpurun_out/prof/mv8_results.db 8 gpurun_out/prof/mv20_results.db 20 > gpurun_out/r03_memvla_infer_kernel_stats.txt
head -40 gpurun_out/r03_memvla_infer_kernel_stats.txt | cut -c1-170
cd /tmp
for n in 2 5; do SKIP_INFER=1 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o pi$n -- python $R/scripts/pi0_bench.py $n > $R/gpurun_out/r03_pi0_$n.log 2>&1; done
cd $R
python profiles/rocpd_stats.py --per-step gpurun_out/prof/pi2_results.db 4 gpurun_out/prof/pi5_results.db 7 > gpurun_out/r03_pi0_train_per_step_kernel_stats.txt
head -40 gpurun_out/r03_pi0_train_per_step_kernel_stats.txt | cut -c1-170
rm -rf gpurun_out/prof
