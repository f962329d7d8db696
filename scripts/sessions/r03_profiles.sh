#!/bin/bash
# round-3 evidence, run on the GPU box from the repo root (outputs under gpurun_out/, copied to profiles/ afterwards):
#   gpurun -- 'bash scripts/r03_profiles.sh [bench|trace|infer|smi|pmc ...]'
set -x
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/prof
what="${@:-bench trace infer smi pmc}"
for w in $what; do case $w in
bench)   # the full default line (driver contract): headline, roofline, recipe figure, p50, cpu_baseline, secondary
  python bench.py > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err; tail -c 1500 gpurun_out/r03_bench.json ;;
trace)   # exact per-step kernel table: difference of two kernel traces of 3 and 9 steps
  cd /tmp; export TMPDIR=/tmp
  for n in 3 9; do rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o tr$n -- python $R/bench.py --steps $n --warmup 1 --no-cpu-baseline --no-latency --no-secondary --no-recipe > $R/gpurun_out/r03_trace_$n.log 2>&1; done
  cd $R
  python profiles/rocpd_stats.py --per-step gpurun_out/prof/tr3_results.db 4 gpurun_out/prof/tr9_results.db 10 > gpurun_out/r03_train_per_step_kernel_stats.txt
  python profiles/rocpd_stats.py gpurun_out/prof/tr9_results.db > gpurun_out/r03_bench_kernel_stats.txt
  grep "^{" gpurun_out/r03_trace_9.log > gpurun_out/r03_bench_profiled.json
  head -30 gpurun_out/r03_train_per_step_kernel_stats.txt | cut -c1-170 ;;
infer)   # per-request kernel table of action inference: difference of traces with 10 and 30 requests
  cd /tmp; export TMPDIR=/tmp
  for n in 10 30; do REQS=$n rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o in$n -- python $R/scripts/infer_bench.py eager > $R/gpurun_out/r03_infer_$n.log 2>&1; done
  cd $R
  python profiles/rocpd_stats.py --per-step gpurun_out/prof/in10_results.db 10 gpurun_out/prof/in30_results.db 30 > gpurun_out/r03_infer_kernel_stats.txt
  tail -1 gpurun_out/r03_infer_30.log; head -16 gpurun_out/r03_infer_kernel_stats.txt | cut -c1-170 ;;
smi)     # clocks / power / power cap while the timed steps run
  ( echo "# rocm-smi --showmaxpower --showperflevel (once), then --showclocks --showpower once per second from second 5 on while 'python bench.py --steps 50 --warmup 2 --no-cpu-baseline --no-latency --no-secondary --no-recipe' runs"
    rocm-smi --showmaxpower --showperflevel 2>&1 | grep -v "^$" ) > gpurun_out/r03_smi_during_bench.txt
  python bench.py --steps 50 --warmup 2 --no-cpu-baseline --no-latency --no-secondary --no-recipe > gpurun_out/r03_smi_bench.json 2>/dev/null &
  BP=$!
  sleep 5
  for i in $(seq 1 24); do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket Graphics|Package Power" | sed -e 's/.*sclk clock level[^(]*//' -e 's/.*(W): //' | tr '\n' ' ' >> gpurun_out/r03_smi_during_bench.txt
    echo >> gpurun_out/r03_smi_during_bench.txt; sleep 1
  done
  wait $BP; cat gpurun_out/r03_smi_during_bench.txt | tail -20 ;;
pmc)
  bash scripts/pmc_passes.sh ;;
esac; done
rm -rf gpurun_out/prof gpurun_out/pmc3/*.db
