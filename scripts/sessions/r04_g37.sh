#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
for v in 0 3 0 3; do echo -n "pi0 DXA_WGRAD_STREAM=$v: "; DXA_WGRAD_STREAM=$v SKIP_INFER=1 timeout 300 python scripts/pi0_bench.py 2>/dev/null | grep "^{" | cut -c1-160; done | tee gpurun_out/r04_wgrad_stream_pi0.txt
timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['value'], d['ms_per_step'], 'frac', r['frac'], 'p50', d['p50_action_inference_ms'], d['reference_recipe_8x_accum2'])"
