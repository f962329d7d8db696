#!/bin/bash
# Round 6: is the process_frame tail the cgroup's CPU-bandwidth throttle (100 ms CFS periods) hit by OpenMP workers spinning after the
# host-side torch.stack of the frames?   gpurun --timeout 600 -- 'bash scripts/sessions/r06_pf2.sh'
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
cd /tmp; export TMPDIR=/tmp
F='^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP version\|amdgpu.ids\|CLIPImageProcessor'
timeout 300 python $R/scripts/pf_trace.py 24 plain threads1 plain threads1 back2back 2>&1 | grep -v "$F" > $R/gpurun_out/r06_pf_threads.txt
cut -c1-420 $R/gpurun_out/r06_pf_threads.txt
OMP_WAIT_POLICY=PASSIVE timeout 300 python $R/scripts/pf_trace.py 24 plain 2>&1 | grep -v "$F" > $R/gpurun_out/r06_pf_passive.txt
cut -c1-420 $R/gpurun_out/r06_pf_passive.txt
