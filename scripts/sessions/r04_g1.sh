#!/bin/bash
# round 4, GPU call 1: w4 kernel correctness + timing vs ping-pong vs library; Tensile kernel names of the library on the step's shapes
set -x
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 600 python scripts/w4_check.py check time lib > gpurun_out/r04_w4_check.log 2>&1; echo "rc $?" >> gpurun_out/r04_w4_check.log
tail -40 gpurun_out/r04_w4_check.log
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_lib -o lib -- python $R/scripts/gemm_bench.py --lib fwd,dX,dW > $R/gpurun_out/r04_lib_trace.log 2>&1
cd $R
python profiles/rocpd_stats.py --names gpurun_out/prof_lib/lib_results.db > gpurun_out/r04_hipblaslt_kernel_names.txt 2>&1
head -40 gpurun_out/r04_hipblaslt_kernel_names.txt | cut -c1-400
rm -rf gpurun_out/prof_lib
timeout 500 python -m pytest tests/test_depth28_gpu.py -x -q -s > gpurun_out/r04_depth28.log 2>&1; tail -15 gpurun_out/r04_depth28.log
