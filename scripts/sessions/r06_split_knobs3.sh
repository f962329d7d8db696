#!/bin/bash
# does DXA_SPLIT_MIN_PIECE=32 (the training optimum) cost the serving paths anything?  few-row products per shape, the CogACT request (2 views and 1 view), MemVLA's frame
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06_split_knobs
O=gpurun_out/r06_split_knobs; rm -f $O/serving.txt
for cfg in "DXA_X=0" "DXA_SPLIT_MIN_PIECE=32"; do
  for rows in 543 287; do
    echo "# ROWS=$rows $cfg" | tee -a $O/serving.txt
    env $cfg ROWS=$rows timeout 300 python scripts/prefill_gemm_bench.py 2>&1 | grep -v amdgpu.ids | tee -a $O/serving.txt
  done
done
for i in 1 2; do
  for cfg in "DXA_X=0" "DXA_SPLIT_MIN_PIECE=32"; do
    echo "request 2 views $cfg: $(env $cfg timeout 300 python scripts/infer_bench.py graph 2>&1 | tail -1 | cut -c1-160)" | tee -a $O/serving.txt
    echo "request 1 view  $cfg: $(env $cfg VIEWS=1 timeout 300 python scripts/infer_bench.py graph 2>&1 | tail -1 | cut -c1-160)" | tee -a $O/serving.txt
  done
done
