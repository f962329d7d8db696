#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_zz_dp_gpu.py tests/test_memvla_gpu.py -m gpu -x -q 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl" | tail -4
for v in 0 3; do echo -n "memvla DXA_WGRAD_STREAM=$v: "; DXA_WGRAD_STREAM=$v SKIP_INFER=1 timeout 300 python scripts/memvla_bench.py 2>/dev/null | grep "^{" | cut -c1-200; done | tee gpurun_out/r04_wgrad_stream_memvla.txt
