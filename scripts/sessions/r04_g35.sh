#!/bin/bash
# A/B/C on one box: dW side stream off / fp32 head only / + bias column sums, three rounds of 12 steps each
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
for rep in 1 2 3; do for v in 0 2 3; do
  echo -n "rep $rep DXA_WGRAD_STREAM=$v: "
  DXA_WGRAD_STREAM=$v timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-secondary --no-latency --no-recipe 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['value'], d['ms_per_step'], 'frac', r['frac'], {k:v['avg_launch_us'] for k,v in r['by_layout'].items()})"
done; done 2>&1 | tee gpurun_out/r04_wgrad_stream_abc.txt
