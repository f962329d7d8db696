#!/bin/bash
# round 6: after the pinned-upload changes — parity tests of the touched models, MemVLA step + per-frame inference
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_after_uploads; mkdir -p $O
timeout 1800 python -m pytest tests/test_memvla_gpu.py tests/test_pi0_gpu.py tests/test_lm_gpu.py tests/test_lm_real_gpu.py tests/test_hybrid_gpu.py -q -x 2>&1 | tail -3 | tee $O/tests.txt
for i in 1 2; do timeout 900 python scripts/memvla_bench.py 10 2>&1 | tail -1 | cut -c1-600 | tee -a $O/memvla.txt; done
