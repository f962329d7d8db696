#!/bin/bash
# the persistent DiT sampler's phase cost taken apart: exchange-protocol probe + the kernel's own ablation switches
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
true
for d in 0 1 2 3 4; do echo "== DXA_DIT_DBG=$d"; DXA_DIT_DBG=$d timeout 120 python scripts/dit_fused_bench.py 2>&1 | grep -v amdgpu.ids | grep "^fused:"; done | tee gpurun_out/r04_dit_dbg.txt
for g in 96 144; do echo "== DXA_DIT_GRID=$g"; DXA_DIT_GRID=$g timeout 120 python scripts/dit_fused_bench.py 2>&1 | grep -v amdgpu.ids | grep "^fused:"; done | tee -a gpurun_out/r04_dit_dbg.txt
