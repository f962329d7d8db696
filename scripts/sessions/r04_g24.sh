#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
ROWS=512,31,543 timeout 300 python scripts/prefill_gemm_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_prefill_rowsplit.txt
