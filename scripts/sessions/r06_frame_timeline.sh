#!/bin/bash
# round 6: where one MemVLA frame (and one CogACT request, eager) waits: step_timeline between two ends of the one-launch sampler
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/r06_frame_timeline; mkdir -p $O $R/gpurun_out/prof
export TMPDIR=/tmp; cd /tmp
FRAMES=10 SKIP_TRAIN=1 timeout 900 rocprofv3 --kernel-trace -d $R/gpurun_out/prof -o mf -- python $R/scripts/memvla_bench.py > $O/memvla_run.log 2>&1
cd $R
python scripts/step_timeline.py gpurun_out/prof/mf_results.db dit_sample_bf16_k > $O/memvla_frame_timeline.txt 2>&1
sed -n 2,8p $O/memvla_frame_timeline.txt | cut -c1-170; grep -A 12 "idle after kernel" $O/memvla_frame_timeline.txt | cut -c1-120; grep -A 8 "per 10 ms window" $O/memvla_frame_timeline.txt; grep -A 12 "25 largest gaps" $O/memvla_frame_timeline.txt | cut -c1-170
rm -rf gpurun_out/prof
