#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_memvla_gpu.py -q -s -k "real_size" 2>&1 | grep -v amdgpu.ids | tail -30 | tee gpurun_out/r04_memvla_real_bf16_fixture.txt
