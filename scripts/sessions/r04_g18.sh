#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
(for d in 0 5 6; do echo "== DXA_DIT_DBG=$d (5: weight loads return zeros without traffic, 6: activation loads)"; DXA_DIT_DBG=$d DXA_LIB=_abl/lib_ditstamp0.so timeout 120 python scripts/dit_stamps.py 2>&1 | grep -v amdgpu.ids; done) | tee gpurun_out/r04_dit_stamps_noW_noA.txt
