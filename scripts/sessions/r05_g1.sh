#!/bin/bash
# round 5, GPU session 1: the new 2-rank model-step test + the tests the advisor fixes touch, the bench line with its new fields,
# one profiled step laid out per stream (scripts/step_timeline.py)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/prof
timeout 600 python -m pytest tests/test_zz_dp2_gpu.py tests/test_zz_dp_gpu.py tests/test_hf_trainer_gpu.py -x -q -s > gpurun_out/r05_g1_tests.txt 2>&1
tail -25 gpurun_out/r05_g1_tests.txt
timeout 900 python bench.py --steps 8 --warmup 3 > gpurun_out/r05_g1_bench.json 2> gpurun_out/r05_g1_bench.err
tail -c 3000 gpurun_out/r05_g1_bench.json; tail -5 gpurun_out/r05_g1_bench.err
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/prof -o tl -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-latency --no-secondary --no-recipe > $R/gpurun_out/r05_g1_tl.log 2>&1
cd $R
python scripts/step_timeline.py gpurun_out/prof/tl_results.db > gpurun_out/r05_step_timeline.txt 2>&1
head -70 gpurun_out/r05_step_timeline.txt | cut -c1-220
rm -f gpurun_out/prof/*.db
