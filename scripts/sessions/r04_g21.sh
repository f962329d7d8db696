#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -s -k "dit_sample_bf16" 2>&1 | grep -v amdgpu.ids | grep "bf16-operand\|passed\|failed\|Error" | tail -12
timeout 900 python -m pytest tests/test_parity_gpu.py -q -x -k "sampler" 2>&1 | grep -B 30 "short test summary" | grep "^E\|assert" | head -20
