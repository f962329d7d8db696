#!/bin/bash
# workgroups of the one-launch bf16 DDIM sampler (DXA_DIT_GRID caps the default) against the request's p50, one box
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06_dit_grid
O=gpurun_out/r06_dit_grid; rm -f $O/*.txt
for g in 0 96 128 160 224; do
  echo "DXA_DIT_GRID=$g: $(DXA_DIT_GRID=$g timeout 300 python scripts/infer_bench.py graph 2>&1 | tail -1 | cut -c1-120)" | tee -a $O/ab.txt
done
echo "DXA_DIT_GRID=0: $(DXA_DIT_GRID=0 timeout 300 python scripts/infer_bench.py graph 2>&1 | tail -1 | cut -c1-120)" | tee -a $O/ab.txt
