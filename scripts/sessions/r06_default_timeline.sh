#!/bin/bash
# round 6: timeline of the DEFAULT headline step (no data-parallel emulation leg in the trace): where the main stream waits
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r06_default_timeline; mkdir -p $O $R/gpurun_out/prof
export TMPDIR=/tmp; cd /tmp
timeout 900 rocprofv3 --kernel-trace -d $R/gpurun_out/prof -o cgd -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-latency --no-secondary --no-recipe --no-dp-emulation > $O/cogact_run.log 2>&1
cd $R
python scripts/step_timeline.py gpurun_out/prof/cgd_results.db > $O/step_timeline.txt 2>&1
sed -n 2,10p $O/step_timeline.txt | cut -c1-200
grep -n "idle after kernel" -A12 $O/step_timeline.txt | cut -c1-150
grep -n "largest gaps" -A26 $O/step_timeline.txt | cut -c1-260
tail -2 $O/cogact_run.log | cut -c1-300
rm -rf gpurun_out/prof
