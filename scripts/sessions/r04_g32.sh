#!/bin/bash
# activation recompute + coalesced accumulation groups: their GPU tests, the default bench line (recipe figure both ways) and the
# headline step with recompute on
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_recompute_gpu.py tests/test_hf_trainer_gpu.py -m gpu -q -s 2>&1 | grep -v "^$" | tail -60 | tee gpurun_out/r04_recompute_coalesce_tests.txt
timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/r04_bench_coalesce.log 2>&1; grep "^{" gpurun_out/r04_bench_coalesce.log > gpurun_out/r04_bench_coalesce.json; cut -c1-1500 gpurun_out/r04_bench_coalesce.json; tail -5 gpurun_out/r04_bench_coalesce.log | cut -c1-300
timeout 600 python bench.py --recompute --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-latency --no-recipe > gpurun_out/r04_bench_recompute.log 2>&1; grep "^{" gpurun_out/r04_bench_recompute.log > gpurun_out/r04_bench_recompute.json; cut -c1-1500 gpurun_out/r04_bench_recompute.json; tail -5 gpurun_out/r04_bench_recompute.log | cut -c1-300
