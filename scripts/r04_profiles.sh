#!/bin/bash
# round-4 evidence, run on the GPU box from the repo root (outputs under gpurun_out/, copied to profiles/ afterwards):
#   gpurun -- 'bash scripts/r04_profiles.sh [recipe|trace|infer|contention ...]'
set -x
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/prof
what="${@:-recipe}"
for w in $what; do case $w in
recipe)  # exact per-optimizer-step kernel table of the reference recipe (8 episodes x 2 accumulation steps): difference of 3- and 9-step traces
  cd /tmp; export TMPDIR=/tmp
  for n in 3 9; do rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o rc$n -- python $R/bench.py --batch 16 --accum 2 --steps $n --warmup 1 --no-cpu-baseline --no-latency --no-secondary --no-recipe > $R/gpurun_out/r04_recipe_$n.log 2>&1; done
  cd $R
  python profiles/rocpd_stats.py --per-step gpurun_out/prof/rc3_results.db 4 gpurun_out/prof/rc9_results.db 10 > gpurun_out/r04_recipe_per_step_kernel_stats.txt
  grep "^{" gpurun_out/r04_recipe_9.log | cut -c1-300
  head -45 gpurun_out/r04_recipe_per_step_kernel_stats.txt | cut -c1-170 ;;
trace)   # exact per-step kernel table of the headline (16 episodes, no accumulation)
  cd /tmp; export TMPDIR=/tmp
  for n in 3 9; do rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o tr$n -- python $R/bench.py --steps $n --warmup 1 --no-cpu-baseline --no-latency --no-secondary --no-recipe > $R/gpurun_out/r04_trace_$n.log 2>&1; done
  cd $R
  python profiles/rocpd_stats.py --per-step gpurun_out/prof/tr3_results.db 4 gpurun_out/prof/tr9_results.db 10 > gpurun_out/r04_train_per_step_kernel_stats.txt
  python profiles/rocpd_stats.py gpurun_out/prof/tr9_results.db > gpurun_out/r04_bench_kernel_stats.txt
  grep "^{" gpurun_out/r04_trace_9.log > gpurun_out/r04_bench_profiled.json
  head -40 gpurun_out/r04_train_per_step_kernel_stats.txt | cut -c1-170 ;;
infer)   # per-request kernel table of action inference: difference of traces with 10 and 30 requests
  cd /tmp; export TMPDIR=/tmp
  for n in 10 30; do REQS=$n rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o in$n -- python $R/scripts/infer_bench.py eager > $R/gpurun_out/r04_infer_$n.log 2>&1; done
  cd $R
  python profiles/rocpd_stats.py --per-step gpurun_out/prof/in10_results.db 10 gpurun_out/prof/in30_results.db 30 > gpurun_out/r04_infer_kernel_stats.txt
  tail -1 gpurun_out/r04_infer_30.log; head -24 gpurun_out/r04_infer_kernel_stats.txt | cut -c1-170 ;;
esac; done
rm -rf gpurun_out/prof
