#!/bin/bash
# round 6: pi0 after "host side first" (masks / positions computed and uploaded from pinned memory before the tower is launched):
# parity tests, the fine-tune step + per-sample inference, and the step timeline (idle per 10 ms window)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r06_pi0_uploads; mkdir -p $O $R/gpurun_out/prof
timeout 1200 python -m pytest tests/test_pi0_gpu.py -q -x 2>&1 | tail -3 | tee $O/tests.txt
for i in 1 2 3; do timeout 600 python scripts/pi0_bench.py 10 16 2>&1 | tail -1 | cut -c1-400 | tee -a $O/bench.txt; done
export TMPDIR=/tmp; cd /tmp
SKIP_INFER=1 timeout 900 rocprofv3 --kernel-trace -d $R/gpurun_out/prof -o pi -- python $R/scripts/pi0_bench.py 3 16 > $O/pi0_run.log 2>&1
cd $R
python scripts/step_timeline.py gpurun_out/prof/pi_results.db > $O/pi_step_timeline.txt 2>&1; sed -n 2,3p $O/pi_step_timeline.txt | cut -c1-150; grep -A 30 "per 10 ms window" $O/pi_step_timeline.txt | grep -v "idle   0.0[0-9] ms"
rm -rf gpurun_out/prof
