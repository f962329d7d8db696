#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r04_gpu_tests.log 2>&1; tail -8 gpurun_out/r04_gpu_tests.log
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-latency --no-secondary --no-recipe 2>/dev/null | grep "^{" | cut -c1-160
SKIP_INFER=1 python scripts/memvla_bench.py 4 2>&1 | tail -1
