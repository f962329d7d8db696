#!/bin/bash
# what the live roofline's HIP events cost the step they measure: every launch timed / every 5th / none
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
for st in 1 5 1000000; do
  echo "== --profile-stride $st"
  timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-secondary --no-latency --no-recipe --profile-stride $st 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['value'], d['ms_per_step'], 'frac', r['frac'], 'timed', r['launches_timed'], 'of', r['launches'], {k:(v['avg_launch_us'],v['achieved']) for k,v in r['by_layout'].items()})"
done 2>&1 | tee gpurun_out/r04_profile_stride.txt
