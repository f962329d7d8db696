#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_parity_gpu.py tests/test_zz_dp_gpu.py tests/test_hf_trainer_gpu.py -x -q > gpurun_out/r04_t3.log 2>&1; tail -25 gpurun_out/r04_t3.log | cut -c1-200
python bench.py --batch 16 --accum 2 --steps 8 --warmup 2 --no-cpu-baseline --no-latency --no-secondary --no-recipe 2>/dev/null | grep "^{" | cut -c1-200
DXA_NO_ACCUM_MERGE=1 python bench.py --batch 16 --accum 2 --steps 8 --warmup 2 --no-cpu-baseline --no-latency --no-secondary --no-recipe 2>/dev/null | grep "^{" | cut -c1-200
