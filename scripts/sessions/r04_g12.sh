#!/bin/bash
# the layer's products at the reference recipe's micro-batch (M = 2296) vs the headline's (M = 4592): ping-pong kernel vs library
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
W4_PP_ONLY=1 W4_M=2296 python scripts/w4_check.py time lib 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_pp_m2296.txt
W4_PP_ONLY=1 python scripts/w4_check.py time lib 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_pp_m4592.txt
