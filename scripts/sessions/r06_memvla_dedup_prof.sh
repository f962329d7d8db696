#!/bin/bash
# round 6: per-step kernel tables of the MemVLA fine-tune step, perceptual tokens repeated (reference layout, MODES has 1) vs distinct (0)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r06_memvla_dedup; mkdir -p $O $R/gpurun_out/prof
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "attention_small" -x 2>&1 | tail -3
for mode in ${MODES:-1 0}; do
  cd /tmp
  for n in 1 3; do DXA_MEMVLA_PER_REPEAT=$mode SKIP_INFER=1 timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o mm${mode}_$n -- python $R/scripts/memvla_bench.py $n > $O/prof_run${mode}_$n.log 2>&1; done
  cd $R
  python profiles/rocpd_stats.py --per-step gpurun_out/prof/mm${mode}_1_results.db 1 gpurun_out/prof/mm${mode}_3_results.db 3 > $O/per_step_mode$mode.txt 2>&1
  head -1 $O/per_step_mode$mode.txt; grep "attn_bwd_small\|attn_fwd_small" $O/per_step_mode$mode.txt | cut -c1-170
done
rm -rf gpurun_out/prof
