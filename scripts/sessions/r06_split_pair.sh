#!/bin/bash
# both operand splits of a bf16x3 product in one launch (dxa_split3_pair): tests, then CogACT / MemVLA steps alternating with DXA_NO_SPLIT_PAIR=1 in one box
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06_split_pair
O=gpurun_out/r06_split_pair; rm -f $O/*.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_memvla_gpu.py -q -m gpu -k "split3 or deferred or memvla" > $O/tests.txt 2>&1; grep -E "passed|failed" $O/tests.txt | tail -1; grep -E "^FAILED|^E  " $O/tests.txt | head -10
for i in 1 2 3; do
  SKIP_INFER=1 timeout 600 python scripts/memvla_bench.py 3 2>&1 | tail -1 | cut -c1-110 | sed 's/^/pair    /' | tee -a $O/memvla.txt
  DXA_NO_SPLIT_PAIR=1 SKIP_INFER=1 timeout 600 python scripts/memvla_bench.py 3 2>&1 | tail -1 | cut -c1-110 | sed 's/^/single  /' | tee -a $O/memvla.txt
done
for i in 1 2 3; do
  for m in 0 1; do
    DXA_NO_SPLIT_PAIR=$m timeout 300 python bench.py --steps 12 --warmup 3 --no-recipe --no-cpu-baseline --no-secondary --no-latency --no-dp-emulation 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('cogact NO_SPLIT_PAIR=$m ms/step', d['ms_per_step'])" | tee -a $O/cogact.txt
  done
done
