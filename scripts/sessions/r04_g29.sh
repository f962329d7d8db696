#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee gpurun_out/r04_gpu_tests_pp3.txt
timeout 600 python scripts/infer_bench.py eager 2>&1 | grep "^eager" | tee gpurun_out/r04_infer_p50_pp3.txt
DXA_GEMM_PP3=0 timeout 600 python scripts/infer_bench.py eager 2>&1 | grep "^eager" | tee -a gpurun_out/r04_infer_p50_pp3.txt
