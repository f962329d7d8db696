#!/bin/bash
set -x
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q 2>&1 | tail -3
timeout 900 python -m pytest tests/test_memvla_gpu.py tests/test_parity_gpu.py tests/test_small_heads_gpu.py -x -q 2>&1 | tail -3
SKIP_TRAIN=1 python scripts/memvla_bench.py 2>&1 | tail -1
DXA_SKINNY_TARGET=0 SKIP_TRAIN=1 python scripts/memvla_bench.py 2>&1 | tail -1
python scripts/infer_bench.py eager 2>&1 | tail -1
