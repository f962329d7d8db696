#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "dit_" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_parity_gpu.py -q -x -k "sampler or inference" 2>&1 | tail -2
timeout 300 python scripts/sampler_bf16_check.py 2>&1 | grep -v amdgpu.ids | tail -3 | tee gpurun_out/r04_sampler_final.txt
timeout 120 python scripts/dit_fused_bench.py 2>&1 | grep "^fused" | tee -a gpurun_out/r04_sampler_final.txt
timeout 600 python scripts/infer_bench.py eager 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a gpurun_out/r04_sampler_final.txt
