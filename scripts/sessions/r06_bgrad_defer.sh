#!/bin/bash
# deferred bias-gradient column sums + cached arena views: MemVLA parity suite, then the step (3 runs) and CogACT (2 runs)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06_bgrad
O=gpurun_out/r06_bgrad
timeout 900 python -m pytest tests/test_memvla_gpu.py tests/test_parity_gpu.py tests/test_zz_dp_gpu.py -x -q -m gpu > $O/tests.txt 2>&1; tail -3 $O/tests.txt
for i in 1 2 3; do SKIP_INFER=1 timeout 600 python scripts/memvla_bench.py 3 2>&1 | tail -1 | cut -c1-200 | tee -a $O/memvla.txt; done
for i in 1 2; do
  timeout 300 python bench.py --steps 12 --warmup 3 --no-recipe --no-cpu-baseline --no-secondary --no-latency --no-dp-emulation 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('cogact ms/step', d['ms_per_step'])" | tee -a $O/cogact.txt
done
