#!/bin/bash
# round 6: where each fine-tune step waits (scripts/step_timeline.py: idle per 10 ms window) — pi0 and the headline CogACT step
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r06_timelines; mkdir -p $O $R/gpurun_out/prof
export TMPDIR=/tmp; cd /tmp
SKIP_INFER=1 timeout 900 rocprofv3 --kernel-trace -d $R/gpurun_out/prof -o pi -- python $R/scripts/pi0_bench.py 3 16 > $O/pi0_run.log 2>&1
timeout 900 rocprofv3 --kernel-trace -d $R/gpurun_out/prof -o cg -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-latency --no-secondary --no-recipe > $O/cogact_run.log 2>&1
cd $R
for w in pi cg; do python scripts/step_timeline.py gpurun_out/prof/${w}_results.db > $O/${w}_step_timeline.txt 2>&1; echo "== $w"; sed -n 2,3p $O/${w}_step_timeline.txt | cut -c1-150; grep -A 30 "per 10 ms window" $O/${w}_step_timeline.txt | grep -v "idle   0.0[0-9] ms"; done
rm -rf gpurun_out/prof
