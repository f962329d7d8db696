#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_memvla_gpu.py tests/test_hf_trainer_gpu.py -x -q > gpurun_out/r04_t2.log 2>&1; tail -30 gpurun_out/r04_t2.log
python scripts/torch_op_census.py memvla 2>&1 | grep -v amdgpu.ids | head -50
SKIP_INFER=1 python scripts/memvla_bench.py 4 2>&1 | tail -1
