#!/bin/bash
# round 6: MemVLA fine-tune step, the two small host->device uploads at the end of the decoder forward from pageable memory (round 5,
# DXA_MEMVLA_PAGEABLE_UPLOAD=1) vs with the plan / from pinned memory; one box, alternating, then the step timeline of the new code
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r06_memvla_upload; mkdir -p $O $R/gpurun_out/prof; rm -f $O/ab_*.txt
for i in 1 2 3; do
  DXA_MEMVLA_PAGEABLE_UPLOAD=1 SKIP_INFER=1 timeout 600 python scripts/memvla_bench.py 10 2>&1 | tail -1 >> $O/ab_pageable.txt
  SKIP_INFER=1 timeout 600 python scripts/memvla_bench.py 10 2>&1 | tail -1 >> $O/ab_pinned.txt
done
for f in ab_pageable ab_pinned; do echo $f; cut -c1-110 $O/$f.txt; done
export TMPDIR=/tmp; cd /tmp
SKIP_INFER=1 timeout 900 rocprofv3 --kernel-trace -d $R/gpurun_out/prof -o tl -- python $R/scripts/memvla_bench.py 3 > $O/timeline_run.log 2>&1
cd $R
python scripts/step_timeline.py gpurun_out/prof/tl_results.db > $O/step_timeline.txt 2>&1; grep -A 40 "per 10 ms window" $O/step_timeline.txt
rm -rf gpurun_out/prof
