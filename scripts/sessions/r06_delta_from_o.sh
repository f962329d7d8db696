#!/bin/bash
# attn_bwd_small_f32_k: delta from the forward's output instead of a first pass over every (query chunk, key chunk); parity + step A/B
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06_delta
O=gpurun_out/r06_delta
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attention" > $O/kernels.txt 2>&1; tail -3 $O/kernels.txt
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_memvla_gpu.py -x -q -m gpu > $O/parity.txt 2>&1; tail -3 $O/parity.txt
for i in 1 2; do
  timeout 300 python bench.py --steps 12 --warmup 3 --no-recipe --no-cpu-baseline --no-secondary --no-latency --no-dp-emulation 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('cogact ms/step', d['ms_per_step'])" | tee -a $O/ab.txt
done
timeout 600 python scripts/memvla_bench.py 3 2>&1 | tail -4 | tee -a $O/ab.txt
