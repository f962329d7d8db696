#!/bin/bash
# how sensitive is the 5-step k_proj-bias movement (a near-cancelling gradient under AdamW) to last-bit changes upstream?
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06_kbias
O=gpurun_out/r06_kbias; rm -f $O/*.txt
T=tests/test_realwidth_gpu.py::test_bf16_five_optimizer_steps_track_the_reference_under_autocast
for cfg in "DXA_X=0" "DXA_ATTN_NO_SMALL_BWD=1" "DXA_WGRAD_STREAM=0" "DXA_GEMM_NO_T128=1"; do
  echo "# $cfg" | tee -a $O/kbias.txt
  env $cfg timeout 600 python -m pytest $T -q -m gpu -s 2>&1 | grep -E "k_proj.bias|losses|norms  |passed|failed" | cut -c1-140 | tee -a $O/kbias.txt
done
