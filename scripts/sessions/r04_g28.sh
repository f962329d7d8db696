#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 300 python scripts/pp3_check.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_pp3_hash_pp3.txt
DXA_GEMM_PP3=0 timeout 300 python scripts/pp3_check.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_pp3_hash_ring.txt
cat gpurun_out/r04_pp3_hash_pp3.txt | cut -c1-160
diff gpurun_out/r04_pp3_hash_pp3.txt gpurun_out/r04_pp3_hash_ring.txt > /dev/null && echo "PP3 == RING bit for bit" || (echo DIFF; diff gpurun_out/r04_pp3_hash_pp3.txt gpurun_out/r04_pp3_hash_ring.txt | head -20)
echo "== pp3"; ROWS=543 timeout 300 python scripts/prefill_gemm_bench.py 2>&1 | grep "^M=" | tee gpurun_out/r04_pp3_time.txt
echo "== ring"; DXA_GEMM_PP3=0 ROWS=543 timeout 300 python scripts/prefill_gemm_bench.py 2>&1 | grep "^M=" | tee -a gpurun_out/r04_pp3_time.txt
