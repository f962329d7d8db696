#!/bin/bash
# round 6: where the MemVLA fine-tune step waits (scripts/step_timeline.py over a 3-step trace)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r06_memvla_timeline2; mkdir -p $O $R/gpurun_out/prof
export TMPDIR=/tmp; cd /tmp
SKIP_INFER=1 timeout 900 rocprofv3 --kernel-trace -d $R/gpurun_out/prof -o tl -- python $R/scripts/memvla_bench.py 3 > $O/timeline_run.log 2>&1
cd $R
python scripts/step_timeline.py gpurun_out/prof/tl_results.db > $O/step_timeline.txt 2>&1; sed -n 2,8p $O/step_timeline.txt | cut -c1-200; grep -A 42 "per 10 ms window" $O/step_timeline.txt; grep -A 14 "idle after kernel" $O/step_timeline.txt | cut -c1-140
rm -rf gpurun_out/prof

