#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
(echo "== bf16-operand sampler"; DXA_LIB=_abl/lib_ditstamp0.so timeout 120 python scripts/dit_sample_stamps.py 2>&1 | grep -v amdgpu.ids | tail -8
for d in 5 6; do echo "== bf16-operand sampler, DXA_DIT_DBG=$d"; DXA_DIT_DBG=$d DXA_LIB=_abl/lib_ditstamp0.so timeout 120 python scripts/dit_sample_stamps.py 2>&1 | grep -v amdgpu.ids | tail -8; done
echo "== exact-fp32 sampler"; DXA_DIT_BF16=0 DXA_LIB=_abl/lib_ditstamp0.so timeout 120 python scripts/dit_sample_stamps.py 2>&1 | grep -v amdgpu.ids | tail -8) | tee gpurun_out/r04_dit_sample_stamps.txt
