#!/bin/bash
set -x
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 900 python scripts/w4_check.py check time lib abl > gpurun_out/r04_w4_dma.log 2>&1; echo "rc $?" >> gpurun_out/r04_w4_dma.log
grep -v "^ok" gpurun_out/r04_w4_dma.log
