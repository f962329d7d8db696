#!/bin/bash
# persistent DiT kernel with branch-free operand loads, balanced K pieces, rewritten attention phase: parity, stamps, timing
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "dit_blocks" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_parity_gpu.py -q -x -k "sampler or inference" 2>&1 | tail -3
DXA_LIB=_abl/lib_ditstamp0.so timeout 120 python scripts/dit_stamps.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_dit_stamps_after.txt
timeout 120 python scripts/dit_fused_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_dit_fused_bench.txt
timeout 120 python scripts/sampler_bench.py 2>&1 | grep -v amdgpu.ids | tail -5 | tee gpurun_out/r04_sampler_bench.txt
