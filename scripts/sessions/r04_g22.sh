#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
(for w in 0 100; do echo "== bf16-operand sampler, 16-counter barrier, workgroup $w"; DXA_LIB=_abl/lib_ditstamp$w.so timeout 120 python scripts/dit_sample_stamps.py 2>&1 | grep -v amdgpu.ids | tail -8; done
echo "== barriers only (DXA_DIT_DBG=1)"; DXA_DIT_DBG=1 timeout 120 python scripts/sampler_bf16_check.py 2>&1 | grep "ms per"
echo "== work only (DXA_DIT_DBG=2)"; DXA_DIT_DBG=2 timeout 120 python scripts/sampler_bf16_check.py 2>&1 | grep "ms per") | tee gpurun_out/r04_dit_sample_stamps_16ctr.txt
