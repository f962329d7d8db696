#!/bin/bash
# Round 5: 8-wave workgroups (128 queries per staged K/V tile) at head_dim 128 / 64 for the forward and dQ kernels, against the
# 4-wave default, inside one box; fingerprints must be equal.   gpurun -- 'bash scripts/attn_nw_probe.sh'
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
out=gpurun_out/r05_attn_nw_probe.txt; : > $out
export ATTN_BENCH_SUMS=1
for e in "DXA_NONE=0" "DXA_ATTN_FWD_NW=8" "DXA_ATTN_DQ_NW=8" "DXA_ATTN_DKV128=2" "DXA_NONE=1"; do
  echo "== attn_bench $e" >> $out
  env $e timeout 300 python scripts/attn_bench.py 2>&1 | grep -v "^ROCm version\|^Hostname\|^Librccl\|^RCCL\|^HIP version\|amdgpu.ids" >> $out
done
echo "== tests DXA_ATTN_FWD_NW=8 DXA_ATTN_DQ_NW=8" >> $out
DXA_ATTN_FWD_NW=8 DXA_ATTN_DQ_NW=8 timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "attention" 2>&1 | tail -2 >> $out
cat $out
