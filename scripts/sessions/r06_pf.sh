#!/bin/bash
# Round 6, verdict item 1a: the every-third-request stall of POST /process_frame, traced.
#   gpurun --timeout 900 -- 'bash scripts/sessions/r06_pf.sh'
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
cd /tmp; export TMPDIR=/tmp
timeout 400 python $R/scripts/pf_trace.py 24 ${PF_MODES:-plain sleep15 keep6 back2back plain} 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP version" > $R/gpurun_out/r06_pf_modes.txt
cut -c1-400 $R/gpurun_out/r06_pf_modes.txt
timeout 400 rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace -f csv -d $R/gpurun_out/pf -o pf -- python $R/scripts/pf_trace.py 18 plain > $R/gpurun_out/r06_pf_traced.txt 2>&1
tail -3 $R/gpurun_out/r06_pf_traced.txt | cut -c1-400
cd $R
python scripts/pf_trace_report.py gpurun_out/pf > gpurun_out/r06_pf_report.txt 2>&1
cat gpurun_out/r06_pf_report.txt | cut -c1-220
rm -rf gpurun_out/pf
