#!/bin/bash
# Round 6: the fixed server loop (numpy stack + pool inside the quota) and whether the TRAINING step sees the CPU quota too.
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
cd /tmp; export TMPDIR=/tmp
F='^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP version\|amdgpu.ids\|CLIPImageProcessor'
timeout 300 python $R/scripts/pf_trace.py 50 plain back2back plain 2>&1 | grep -v "$F" > $R/gpurun_out/r06_pf_fixed.txt
cut -c1-600 $R/gpurun_out/r06_pf_fixed.txt
B="python $R/bench.py --no-cpu-baseline --no-latency --no-secondary --no-recipe --steps 10 --warmup 3"
for v in 0 default 0 default; do
  if [ $v = default ]; then timeout 300 $B 2>&1 | grep "^{" > $R/gpurun_out/r06_bench_threads_$v.json; else DXA_HOST_THREADS=0 timeout 300 $B 2>&1 | grep "^{" > $R/gpurun_out/r06_bench_threads_$v.json; fi
  python - <<PY
import json; d=json.load(open("$R/gpurun_out/r06_bench_threads_$v.json")); print("$v", d["ms_per_step"], d["value"], d["roofline"]["frac"], d["host"])
PY
done
