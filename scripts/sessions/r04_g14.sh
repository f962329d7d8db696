#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_parity_gpu.py -x -q > gpurun_out/r04_t5.log 2>&1; tail -5 gpurun_out/r04_t5.log | cut -c1-200
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-latency --no-secondary --no-recipe 2>/dev/null | grep "^{" | tee gpurun_out/r04_ppr_bench.json | cut -c1-600
DXA_LIB=_abl/lib_old.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-latency --no-secondary --no-recipe 2>/dev/null | grep "^{" | tee gpurun_out/r04_ppr_bench_old.json | cut -c1-600
