#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
{ echo "== round 5 dispatch order (heaviest causal tiles first; dK/dV grid key-tile slowest)"; python scripts/attn_bench.py 2>&1 | grep -v amdgpu
  echo "== rounds 1-4 order (_abl/lib_attn_old.so)"; DXA_LIB=_abl/lib_attn_old.so python scripts/attn_bench.py 2>&1 | grep -v amdgpu
  echo "== round 5 again"; python scripts/attn_bench.py 2>&1 | grep -v amdgpu; } > gpurun_out/r05_attn_order.txt
cat gpurun_out/r05_attn_order.txt
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "attention" 2>&1 | tail -3
B="python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-latency --no-secondary --no-recipe"
for i in 1 2; do for lib in "" "_abl/lib_attn_old.so"; do echo "== bench lib=[$lib]"; DXA_LIB=$lib $B 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['roofline']['frac'])"; done; done > gpurun_out/r05_attn_order_bench.txt 2>&1
cat gpurun_out/r05_attn_order_bench.txt
