#!/bin/bash
# finer sweep of DXA_SPLIT_MIN_PIECE (slabs of 32 per K piece of a split tile; default 16) on the DB-CogACT step, one box, alternating; MemVLA and the request at the best
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06_split_knobs
O=gpurun_out/r06_split_knobs; rm -f $O/ab2.txt
cg() { env "$@" timeout 300 python bench.py --steps 12 --warmup 3 --no-recipe --no-cpu-baseline --no-secondary --no-latency --no-dp-emulation 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'])"; }
for i in 1 2; do
  for cfg in "DXA_X=0" "DXA_SPLIT_MIN_PIECE=24" "DXA_SPLIT_MIN_PIECE=32" "DXA_SPLIT_MIN_PIECE=48" "DXA_SPLIT_MIN_PIECE=64" "DXA_SPLIT_MIN_PIECE=1000"; do
    echo "cogact $cfg  $(cg $cfg)" | tee -a $O/ab2.txt
  done
done
