#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
python scripts/tn_vs_nn_bench.py 2>&1 | grep -v amdgpu > gpurun_out/r05_tn_vs_nn.txt; cat gpurun_out/r05_tn_vs_nn.txt
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-latency --no-secondary --no-recipe"
for gm in 4 2 6 9 18; do echo "== DXA_GEMM_GROUP_M=$gm"; DXA_GEMM_GROUP_M=$gm $B 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print(d['ms_per_step'], r['frac'], {k:(v['avg_launch_us'],v['achieved']) for k,v in r['by_layout'].items()})"; done > gpurun_out/r05_group_m_sweep.txt 2>&1
cat gpurun_out/r05_group_m_sweep.txt
