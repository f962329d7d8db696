#!/bin/bash
# DXA_WGRAD_STREAM=1 (EVERY dW product beside the dX chain) against the default 3 on the headline step, one box, alternating: step time and what the per-launch roofline reads
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06_wgrad1
O=gpurun_out/r06_wgrad1; rm -f $O/*.txt
cg() { env "$@" timeout 300 python bench.py --steps 12 --warmup 3 --no-recipe --no-cpu-baseline --no-secondary --no-latency --no-dp-emulation 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'])"; }
for i in 1 2 3; do
  for cfg in "DXA_WGRAD_STREAM=3" "DXA_WGRAD_STREAM=1"; do echo "cogact $cfg  $(cg $cfg)" | tee -a $O/ab.txt; done
done
