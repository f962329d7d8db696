#!/bin/bash
# round-4 closing evidence on one box: GPU test suite, the default bench line, per-step / per-request kernel tables
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^ROCm version\|^Hostname\|^Librccl" | tail -4 | tee gpurun_out/r04_gpu_tests_final.txt
timeout 900 python bench.py > gpurun_out/r04_bench_default.log 2>&1; grep "^{" gpurun_out/r04_bench_default.log > gpurun_out/r04_bench.json; cut -c1-600 gpurun_out/r04_bench.json
bash scripts/r04_profiles.sh trace infer 2>&1 | tail -70
