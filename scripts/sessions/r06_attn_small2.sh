#!/bin/bash
# register-blocked fp32 attention backward for > 32 queries (attn_bwd_small2_f32_k): parity, bit-identity with the one-element-per-thread kernel, time, MemVLA step A/B
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06_attn_small2
O=gpurun_out/r06_attn_small2; rm -f $O/*.txt
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "attention" > $O/tests.txt 2>&1; grep -E "passed|failed" $O/tests.txt | tail -1
timeout 300 python scripts/attn_small_bwd_crc.py 2>&1 | grep -v amdgpu.ids | tee $O/crc_v2.txt
DXA_ATTN_SMALL_BWD_V1=1 timeout 300 python scripts/attn_small_bwd_crc.py 2>&1 | grep -v amdgpu.ids | tee $O/crc_v1.txt
diff <(sed 's/   .*//' $O/crc_v1.txt) <(sed 's/   .*//' $O/crc_v2.txt) && echo "CRCs identical" | tee -a $O/crc_v2.txt
timeout 600 python -m pytest tests/test_memvla_gpu.py -q -m gpu > $O/memvla_tests.txt 2>&1; grep -E "passed|failed" $O/memvla_tests.txt | tail -1
for i in 1 2; do
  SKIP_INFER=1 timeout 600 python scripts/memvla_bench.py 3 2>&1 | tail -1 | cut -c1-110 | sed 's/^/blocked  /' | tee -a $O/memvla.txt
  DXA_ATTN_SMALL_BWD_V1=1 SKIP_INFER=1 timeout 600 python scripts/memvla_bench.py 3 2>&1 | tail -1 | cut -c1-110 | sed 's/^/v1       /' | tee -a $O/memvla.txt
done
