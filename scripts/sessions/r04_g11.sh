#!/bin/bash
# two-level barrier arrival in the persistent DiT kernels: tests + timings
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "dit or sampler or infer or action" > gpurun_out/r04_t4.log 2>&1; tail -8 gpurun_out/r04_t4.log | cut -c1-200
python scripts/dit_fused_bench.py 2>&1 | tail -6
python scripts/sampler_bench.py 2>&1 | tail -8
python scripts/infer_bench.py 2>&1 | tail -8
