#!/bin/bash
# Round 6: the sharded optimizer step on the MI355X — 2-rank model step (gloo, one GPU), world-1 RCCL sequence, gated RCCL test (skips)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
F='^ROCm version\|^Hostname\|^Librccl\|^RCCL\|^HIP version\|amdgpu.ids'
timeout 1200 python -m pytest tests/test_zz_dp2_gpu.py tests/test_zz_dp_gpu.py tests/test_zz_dp_rccl_gpu.py -m gpu -q -x -s 2>&1 | grep -v "$F" | tail -40 | cut -c1-260 | tee gpurun_out/r06_dp_tests.txt
