#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_zz_dp2_gpu.py tests/test_zz_dp_gpu.py tests/test_parity_gpu.py tests/test_realwidth_gpu.py tests/test_recompute_gpu.py tests/test_hf_trainer_gpu.py -q -x 2>&1 | tail -4
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-latency --no-secondary --no-recipe"
for i in 1 2 3; do for m in 0 1; do echo "== DXA_JOIN_AT_BUCKET=$m"; DXA_JOIN_AT_BUCKET=$m $B 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['roofline']['frac'])"; done; done > gpurun_out/r05_join_ab.txt 2>&1
cat gpurun_out/r05_join_ab.txt
