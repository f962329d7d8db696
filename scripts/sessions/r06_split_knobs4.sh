#!/bin/bash
# with DXA_SPLIT_MIN_PIECE = 32 as the default: the other two split-K knobs on the DB-CogACT step (one box, alternating)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06_split_knobs
O=gpurun_out/r06_split_knobs; rm -f $O/ab4.txt
cg() { env "$@" timeout 300 python bench.py --steps 12 --warmup 3 --no-recipe --no-cpu-baseline --no-secondary --no-latency --no-dp-emulation 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'])"; }
for i in 1 2; do
  for cfg in "DXA_X=0" "DXA_SPLIT_MAX=4" "DXA_SPLIT_MAX=3" "DXA_SPLIT_MIN_NK=128" "DXA_SPLIT_MIN_NK=32" "DXA_SPLIT_MIN_PIECE=40"; do
    echo "cogact $cfg  $(cg $cfg)" | tee -a $O/ab4.txt
  done
done
