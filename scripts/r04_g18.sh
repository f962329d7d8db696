#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
(echo "== 12 distinct weight sets (340 MB)"; DXA_LIB=_abl/lib_ditstamp0.so timeout 120 python scripts/dit_stamps.py 2>&1 | grep -v amdgpu.ids
echo "== every block reads block 0's weights (28 MB, warm)"; DIT_SAME_W=1 DXA_LIB=_abl/lib_ditstamp0.so timeout 120 python scripts/dit_stamps.py 2>&1 | grep -v amdgpu.ids) | tee gpurun_out/r04_dit_stamps_samew.txt
