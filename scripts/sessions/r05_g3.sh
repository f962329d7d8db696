#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 300 python scripts/dp2_debug.py 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo\|Warning" > gpurun_out/r05_dp2_debug.txt; cat gpurun_out/r05_dp2_debug.txt
timeout 900 python -m pytest tests/test_realwidth_gpu.py -q -s -x -k "five_optimizer or twelve" > gpurun_out/r05_g3_tests.txt 2>&1
grep -v "^$\|Warning\|warn\|amdgpu.ids" gpurun_out/r05_g3_tests.txt | tail -80 | cut -c1-250
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_pi0_gpu.py -q -x -k "attention or siglip or pi0" > gpurun_out/r05_g3_attn.txt 2>&1; tail -8 gpurun_out/r05_g3_attn.txt | cut -c1-250
timeout 300 python scripts/pi0_bench.py 3 16 2>&1 | grep "^{" | cut -c1-400
