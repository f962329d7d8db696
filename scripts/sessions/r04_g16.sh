#!/bin/bash
# DXA_PPR=2 (B register sets swap roles per K tile: fragment reads 4/8/4/8 per compute cluster) vs DXA_PPR=1 (_abl/lib_ppr1.so):
# bit-for-bit hash, timing on the layer's shapes, barrier-interval stamps; and the M = 543 prefill products on 192-row ring tiles
# (default) vs forced 256-row ping-pong tiles, plus M = 768 (three full ping-pong row tiles) as the yardstick for a 192-row build
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
python scripts/gemm_hash.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_hash_ppr2.txt
DXA_LIB=_abl/lib_ppr1.so python scripts/gemm_hash.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_hash_ppr1.txt
wc -l gpurun_out/r04_hash_ppr2.txt; diff gpurun_out/r04_hash_ppr2.txt gpurun_out/r04_hash_ppr1.txt && echo "PPR2 == PPR1 bit for bit"
echo "== PPR=2"; W4_PP_ONLY=1 python scripts/w4_check.py time 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_ppr2_time.txt
echo "== PPR=1"; DXA_LIB=_abl/lib_ppr1.so W4_PP_ONLY=1 python scripts/w4_check.py time 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_ppr1_time.txt
echo "== PPR=2 again"; W4_PP_ONLY=1 python scripts/w4_check.py time 2>&1 | grep -v amdgpu.ids | tail -1
DXA_LIB=_abl/lib_stamp2.so python scripts/pp_stamps.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_pp_stamps_ppr2.txt
echo "== prefill products, default dispatch"; ROWS=543,576,768 python scripts/prefill_gemm_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_prefill_default.txt
echo "== prefill products, 256-row ping-pong tiles forced"; DXA_GEMM_RING_AI=4 DXA_GEMM_NO_T128=1 ROWS=543,768 python scripts/prefill_gemm_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_prefill_pp256.txt
