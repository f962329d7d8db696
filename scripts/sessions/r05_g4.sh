#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
{ echo "== round-4 kernel (one wave per row)"; DXA_NORM_BWD_NO_SPLIT=1 python scripts/norm_bench.py 2>&1 | grep rows
  echo "== split kernel, 4 rows per pass (512 threads)"; python scripts/norm_bench.py 2>&1 | grep rows
  echo "== split kernel, 3 rows per pass (384 threads)"; DXA_NORM_BWD_ROWS=3 python scripts/norm_bench.py 2>&1 | grep rows; } > gpurun_out/r05_norm_bwd_split.txt
cat gpurun_out/r05_norm_bwd_split.txt
timeout 600 python -m pytest tests/test_zz_dp2_gpu.py -q -s -x > gpurun_out/r05_g4_dp2.txt 2>&1; grep -v "^$\|Warning\|warn\|amdgpu.ids\|socket.cpp\|Gloo" gpurun_out/r05_g4_dp2.txt | tail -30 | cut -c1-330
timeout 900 python -m pytest tests/test_realwidth_gpu.py -q -s -k "five_optimizer or twelve" > gpurun_out/r05_g4_traj.txt 2>&1; grep -v "^$\|Warning\|warn\|amdgpu.ids" gpurun_out/r05_g4_traj.txt | tail -60 | cut -c1-250
{ echo "== pi0 step, SigLIP head width 72 native"; timeout 300 python scripts/pi0_bench.py 3 16 2>&1 | grep "^{" | cut -c1-200
  echo "== pi0 step, round-3/4 zero-padded copies (DXA_ATTN_PAD72=1)"; DXA_ATTN_PAD72=1 timeout 300 python scripts/pi0_bench.py 3 16 2>&1 | grep "^{" | cut -c1-200
  echo "== native again"; timeout 300 python scripts/pi0_bench.py 3 16 2>&1 | grep "^{" | cut -c1-200; } > gpurun_out/r05_pi0_hd72.txt
cat gpurun_out/r05_pi0_hd72.txt
