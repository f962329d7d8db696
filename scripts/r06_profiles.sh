#!/bin/bash
# Round-5 evidence, run on the GPU box from the repo root (outputs under gpurun_out/, the summaries are copied to profiles/):
#   gpurun -- 'bash scripts/r06_profiles.sh [tests|bench|trace|timeline|infer|pi0|memvla|decode|pmc ...]'
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/prof
what="${@:-tests bench}"
B="python $R/bench.py --no-cpu-baseline --no-latency --no-secondary --no-recipe --no-dp-emulation"   # (the timeline takes the LAST step of the trace: it must be a headline step)
for w in $what; do case $w in
tests)   # the GPU suite
  timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "^ROCm version\|^Hostname\|^Librccl\|^RCCL\|^HIP version" | tail -6 | tee gpurun_out/r06_gpu_tests.txt ;;
bench)   # the default driver-style line
  timeout 1200 python bench.py > gpurun_out/r06_bench_default.log 2>&1; grep "^{" gpurun_out/r06_bench_default.log > gpurun_out/r06_bench.json; cut -c1-700 gpurun_out/r06_bench.json ;;
trace)   # exact per-step kernel table of the headline step: difference of a 4- and a 10-step trace
  cd /tmp; export TMPDIR=/tmp
  for n in 3 9; do rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o tr$n -- $B --steps $n --warmup 1 > $R/gpurun_out/r06_trace_$n.log 2>&1; done
  cd $R
  python profiles/rocpd_stats.py --per-step gpurun_out/prof/tr3_results.db 4 gpurun_out/prof/tr9_results.db 10 > gpurun_out/r06_train_per_step_kernel_stats.txt
  python profiles/rocpd_stats.py gpurun_out/prof/tr9_results.db > gpurun_out/r06_bench_kernel_stats.txt
  python scripts/step_timeline.py gpurun_out/prof/tr9_results.db > gpurun_out/r06_step_timeline.txt 2>&1
  grep "^{" gpurun_out/r06_trace_9.log > gpurun_out/r06_bench_profiled.json
  head -36 gpurun_out/r06_train_per_step_kernel_stats.txt | cut -c1-170 ;;
infer)   # per-request kernel table of action inference: difference of traces with 10 and 30 requests
  cd /tmp; export TMPDIR=/tmp
  for n in 10 30; do REQS=$n rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o in$n -- python $R/scripts/infer_bench.py eager > $R/gpurun_out/r06_infer_$n.log 2>&1; done
  cd $R
  python profiles/rocpd_stats.py --per-step gpurun_out/prof/in10_results.db 10 gpurun_out/prof/in30_results.db 30 > gpurun_out/r06_infer_kernel_stats.txt
  tail -1 gpurun_out/r06_infer_30.log; head -20 gpurun_out/r06_infer_kernel_stats.txt | cut -c1-170 ;;
pi0)     # exact per-step kernel table of the pi0 fine-tune step
  cd /tmp; export TMPDIR=/tmp
  for n in 1 3; do SKIP_INFER=1 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o pi$n -- python $R/scripts/pi0_bench.py $n 16 > $R/gpurun_out/r06_pi0_$n.log 2>&1; done
  cd $R
  python profiles/rocpd_stats.py --per-step gpurun_out/prof/pi1_results.db 3 gpurun_out/prof/pi3_results.db 5 > gpurun_out/r06_pi0_train_per_step_kernel_stats.txt 2>&1
  grep "^{" gpurun_out/r06_pi0_3.log | cut -c1-300; head -26 gpurun_out/r06_pi0_train_per_step_kernel_stats.txt | cut -c1-160 ;;
memvla)  # exact per-step kernel table of the MemVLA fine-tune step (difference of traces with 1 and 3 timed steps)
  cd /tmp; export TMPDIR=/tmp
  for n in 1 3; do SKIP_INFER=1 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o mem$n -- python $R/scripts/memvla_bench.py $n > $R/gpurun_out/r06_memvla_$n.log 2>&1; done
  cd $R
  python profiles/rocpd_stats.py --per-step gpurun_out/prof/mem1_results.db 1 gpurun_out/prof/mem3_results.db 3 > gpurun_out/r06_memvla_train_per_step_kernel_stats.txt 2>&1
  grep "^{" gpurun_out/r06_memvla_3.log | cut -c1-300; head -26 gpurun_out/r06_memvla_train_per_step_kernel_stats.txt | cut -c1-160 ;;
decode)  # per-TOKEN kernel table of the KV-cached greedy decode: difference of traces with 9 and 33 new tokens (11 generations of n tokens each: 8 and 32 single-token passes per generation)
  cd /tmp; export TMPDIR=/tmp
  for n in 9 33; do rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o de$n -- python $R/scripts/decode_bench.py $n > $R/gpurun_out/r06_decode_$n.log 2>&1; done
  cd $R
  python profiles/rocpd_stats.py --per-step gpurun_out/prof/de9_results.db 88 gpurun_out/prof/de33_results.db 352 > gpurun_out/r06_decode_per_token_kernel_stats.txt 2>&1
  tail -3 gpurun_out/r06_decode_33.log; head -24 gpurun_out/r06_decode_per_token_kernel_stats.txt | cut -c1-170 ;;
pmc)     # hardware counters of the dominant kernel: separate --pmc passes (scripts/pmc_passes.sh), stamped with the commit
  PMC_ROUND=r06 bash scripts/pmc_passes.sh > gpurun_out/r06_pmc_passes.log 2>&1
  tail -30 gpurun_out/r06_pmc_passes.log | cut -c1-200 ;;
esac; rm -f gpurun_out/prof/*.db; done
rm -rf gpurun_out/prof
