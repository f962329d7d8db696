#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/prof
timeout 1200 python -m pytest tests -m gpu -q -s -x > gpurun_out/r05_gpu_tests_s.txt 2>&1; tail -4 gpurun_out/r05_gpu_tests_s.txt | cut -c1-200
grep -n "bf16 product vs\|  loss \|  cognition\|  eps_hat\|  gnorm\|  gsamp\|  infer\|infer_samples\|sampler\|bf16 sampler" gpurun_out/r05_gpu_tests_s.txt | cut -c1-220 | head -60
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/prof -o tl -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-latency --no-secondary --no-recipe > $R/gpurun_out/r05_g9_tl.log 2>&1
cd $R
python scripts/step_timeline.py gpurun_out/prof/tl_results.db > gpurun_out/r05_step_timeline.txt 2>&1
sed -n 1,60p gpurun_out/r05_step_timeline.txt | cut -c1-200
rm -f gpurun_out/prof/*.db
