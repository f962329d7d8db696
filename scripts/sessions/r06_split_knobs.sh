#!/bin/bash
# split-K heuristics of the 256-column-tile kernels against the training steps (one box, alternating): DXA_SPLIT_MAX (default 8), DXA_SPLIT_MIN_PIECE (default 16 slabs of 32)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06_split_knobs
O=gpurun_out/r06_split_knobs; rm -f $O/*.txt
cg() { env "$@" timeout 300 python bench.py --steps 12 --warmup 3 --no-recipe --no-cpu-baseline --no-secondary --no-latency --no-dp-emulation 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'])"; }
mv() { env "$@" SKIP_INFER=1 timeout 600 python scripts/memvla_bench.py 3 2>&1 | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for i in 1 2; do
  for cfg in "DXA_X=0" "DXA_SPLIT_MAX=4" "DXA_SPLIT_MAX=2" "DXA_SPLIT_MIN_PIECE=32"; do
    echo "cogact $cfg  $(cg $cfg)   memvla $(mv $cfg)" | tee -a $O/ab.txt
  done
done
