#!/bin/bash
# HIP / HSA runtime knobs against the launch-heavy steps (one box, alternating with the default): kernel-argument placement, interrupt vs polling waits
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06_runtime_env
O=gpurun_out/r06_runtime_env; rm -f $O/*.txt
cg() { env "$@" timeout 300 python bench.py --steps 12 --warmup 3 --no-recipe --no-cpu-baseline --no-secondary --no-latency --no-dp-emulation 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'])"; }
mv() { env "$@" SKIP_INFER=1 timeout 600 python scripts/memvla_bench.py 3 2>&1 | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for i in 1 2; do
  for cfg in "DXA_X=0" "HIP_FORCE_DEV_KERNARG=0" "HIP_FORCE_DEV_KERNARG=1" "HSA_ENABLE_INTERRUPT=0"; do
    echo "cogact $cfg  $(cg $cfg)   memvla $(mv $cfg)" | tee -a $O/ab.txt
  done
done
