#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_hf_trainer_gpu.py tests/test_zz_dp_gpu.py -x -q > gpurun_out/r04_t1.log 2>&1; tail -5 gpurun_out/r04_t1.log
timeout 600 python bench.py --batch 16 --accum 2 --steps 8 --warmup 2 --no-cpu-baseline --no-latency --no-secondary --no-recipe 2>/dev/null | cut -c1-200
