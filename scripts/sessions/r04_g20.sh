#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 300 python scripts/sampler_bf16_check.py 2>&1 | grep -v amdgpu.ids | tail -12 | tee gpurun_out/r04_sampler_bf16_check.txt
