#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
(for s in 4 2 1; do echo "== proj slices $s"; DXA_DIT_PROJ_SLICES=$s timeout 300 python scripts/sampler_bf16_check.py 2>&1 | grep "ms per"; done) | tee gpurun_out/r04_proj_slices.txt
