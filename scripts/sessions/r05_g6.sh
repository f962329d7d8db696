#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
run() { for i in 1 2 3 4 5 6 7 8; do env "$@" DP2_SYNC=0 DP2_N=3 timeout 300 python scripts/dp2_debug3.py 2>&1 | grep "^step" | cut -c1-30 | tr '\n' ' '; echo; done | sort | uniq -c; }
{ echo "== after the fix (attribute writes through the store view reach the store), 8 runs, tracker sum of squares at steps 1 2 3"; run A=1; } > gpurun_out/r05_dp2_after_fix.txt
cat gpurun_out/r05_dp2_after_fix.txt
timeout 900 python -m pytest tests/test_zz_dp2_gpu.py tests/test_zz_dp_gpu.py tests/test_parity_gpu.py -q -s -x > gpurun_out/r05_g6_tests.txt 2>&1; grep -v "^$\|Warning\|warn\|amdgpu.ids\|socket.cpp\|Gloo" gpurun_out/r05_g6_tests.txt | tail -22 | cut -c1-330
timeout 900 python -m pytest tests/test_realwidth_gpu.py -q -s -k "five_optimizer" > gpurun_out/r05_g6_traj.txt 2>&1; grep -v "^$\|Warning\|warn\|amdgpu.ids" gpurun_out/r05_g6_traj.txt | tail -30 | cut -c1-250
