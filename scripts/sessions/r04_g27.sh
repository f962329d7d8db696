#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
(for dk in 2048 1024 512; do echo "== DXA_GEMM_T128_DEEPK=$dk"; DXA_GEMM_T128_DEEPK=$dk REQS=20 timeout 300 python scripts/infer_bench.py eager 2>&1 | grep "^eager"; done
echo "== vit shapes"; for dk in 2048 1024; do echo "DEEPK=$dk"; DXA_GEMM_T128_DEEPK=$dk SHAPES=vit ROWS=514 timeout 120 python scripts/prefill_gemm_bench.py 2>&1 | grep "^M="; done) | tee gpurun_out/r04_t128_deepk.txt
