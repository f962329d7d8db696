#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm" 2>&1 | grep -E "passed|failed" | tail -2
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_realwidth_gpu.py -q -x 2>&1 | grep -E "passed|failed" | tail -2
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-latency --no-secondary --no-recipe 2>/dev/null | grep "^{" | cut -c1-200 | tee gpurun_out/r04_bench_pp3_f32.txt
DXA_GEMM_PP3=0 timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-latency --no-secondary --no-recipe 2>/dev/null | grep "^{" | cut -c1-200 | tee -a gpurun_out/r04_bench_pp3_f32.txt
