#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp; cd $R
rocprofv3 -L > gpurun_out/r04_counters_list.txt 2>&1
grep -o "SQ_[A-Z_0-9]*" gpurun_out/r04_counters_list.txt | sort -u | tr '\n' ' ' | head -c 6000
