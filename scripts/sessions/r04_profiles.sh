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
memvla)  # exact per-step kernel table of the MemVLA fine-tune step (1 + 2 warm-up vs 3 + 2 warm-up steps)
  cd /tmp; export TMPDIR=/tmp
  for n in 1 3; do SKIP_INFER=1 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o mem$n -- python $R/scripts/memvla_bench.py $n > $R/gpurun_out/r04_memvla_$n.log 2>&1; done
  cd $R
  python profiles/rocpd_stats.py --per-step gpurun_out/prof/mem1_results.db 3 gpurun_out/prof/mem3_results.db 5 > gpurun_out/r04_memvla_train_per_step_kernel_stats.txt 2>&1
  grep "^{" gpurun_out/r04_memvla_3.log; head -70 gpurun_out/r04_memvla_train_per_step_kernel_stats.txt | cut -c1-160 ;;
pi0)
  cd /tmp; export TMPDIR=/tmp
  for n in 1 3; do SKIP_INFER=1 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o pi$n -- python $R/scripts/pi0_bench.py $n 16 > $R/gpurun_out/r04_pi0_$n.log 2>&1; done
  cd $R
  python profiles/rocpd_stats.py --per-step gpurun_out/prof/pi1_results.db 3 gpurun_out/prof/pi3_results.db 5 > gpurun_out/r04_pi0_train_per_step_kernel_stats.txt 2>&1
  grep "^{" gpurun_out/r04_pi0_3.log; head -40 gpurun_out/r04_pi0_train_per_step_kernel_stats.txt | cut -c1-160 ;;
contention)   # RCCL kernels of the N > 1 collective sequence sharing the GPU with the backward's GEMM grids, on ONE GPU
  B="python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-latency --no-secondary --no-recipe"
  $B > gpurun_out/r04_cont_plain.json 2>/dev/null
  $B --force-reducer > gpurun_out/r04_cont_sum.json 2>/dev/null
  $B --force-reducer --native-avg > gpurun_out/r04_cont_avg_default.json 2>/dev/null
  NCCL_MAX_NCHANNELS=8 $B --force-reducer --native-avg > gpurun_out/r04_cont_avg_8ch.json 2>/dev/null
  NCCL_MAX_NCHANNELS=4 $B --force-reducer --native-avg > gpurun_out/r04_cont_avg_4ch.json 2>/dev/null
  NCCL_MAX_NCHANNELS=4 $B --force-reducer --native-avg --grad-comm float32 > gpurun_out/r04_cont_avg_4ch_f32.json 2>/dev/null
  python - <<'PY'
import json, glob
out = {}
for f in sorted(glob.glob("gpurun_out/r04_cont_*.json")):
    try:
        d = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
    except Exception as e:
        out[f] = str(e); continue
    bl = d.get("roofline", {}).get("by_layout", {})
    out[f.split("r04_cont_")[1][:-5]] = {"ms_per_step": d["ms_per_step"], "episodes_per_s": d["value"],
        "gemm_avg_launch_us": {k: v.get("avg_launch_us") for k, v in bl.items()}, "gemm_tf": {k: v.get("achieved") for k, v in bl.items()},
        "comm_window_ms_per_step": d.get("comm_window_ms_per_step"), "collectives_per_step": d.get("collectives_per_step"),
        "allreduce_gb_per_step": d.get("allreduce_gb_per_step")}
json.dump(out, open("gpurun_out/r04_reducer_contention.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
  ;;
esac; done
rm -rf gpurun_out/prof
