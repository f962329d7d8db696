#!/bin/bash
# dW products on a side stream beside the dX chain: bit-identity tests + the headline step with and without
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
DXA_WGRAD_STREAM=3 timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_recompute_gpu.py -m gpu -x -q 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl" | tail -4
for v in 2 3 2 3; do
  echo "== DXA_WGRAD_STREAM=$v"
  DXA_WGRAD_STREAM=$v timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-secondary --no-latency --no-recipe 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['value'], d['ms_per_step'], 'loss', d['loss'], 'frac', r['frac'], {k:(v['avg_launch_us'],v['achieved']) for k,v in r['by_layout'].items()})"
done 2>&1 | tee gpurun_out/r04_wgrad_stream_bgrad.txt
