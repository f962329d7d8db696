#!/bin/bash
# MemVLA now GPU-bound (7,229 launches): does the gradient side stream pay now?  DXA_WGRAD_STREAM = 0 (default for MemVLA) / 2 / 3, alternating, one box
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06_memvla_side
O=gpurun_out/r06_memvla_side; rm -f $O/*.txt
for i in 1 2; do
  for m in 0 2 3; do
    DXA_WGRAD_STREAM=$m SKIP_INFER=1 timeout 600 python scripts/memvla_bench.py 3 2>&1 | tail -1 | cut -c1-110 | sed "s/^/side=$m  /" | tee -a $O/ab.txt
  done
done
