#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_hf_trainer_gpu.py tests/test_zz_dp_gpu.py tests/test_kernels_gpu.py -m gpu -x -q -k "adamw or parity or trainer or dp or optim" 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP version" | tail -4
for rep in 1 2; do for v in 0 1; do
  echo -n "rep $rep DXA_ADAMW_SPARSE=$v: "
  DXA_ADAMW_SPARSE=$v timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-secondary --no-latency --no-recipe 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['value'], d['ms_per_step'], 'frac', r['frac'])"
done; done 2>&1 | tee gpurun_out/r04_adamw_sparse.txt
