#!/bin/bash
# reads-in-compute-cluster schedule of the ping-pong kernel (DXA_PPR=1): parity against the unchanged w4 kernel / fp32, timing vs the
# previous schedule (_abl/lib_old.so = -DDXA_PPR=0) and the library, barrier-interval stamps
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
python scripts/w4_check.py check 2>&1 | grep -v amdgpu.ids | tail -45 > gpurun_out/r04_ppr_check.txt; tail -4 gpurun_out/r04_ppr_check.txt; grep -c "bit-identical" gpurun_out/r04_ppr_check.txt; grep BAD gpurun_out/r04_ppr_check.txt | head
echo "== new schedule"; W4_PP_ONLY=1 python scripts/w4_check.py time lib 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_ppr_time_new.txt
echo "== old schedule"; DXA_LIB=_abl/lib_old.so W4_PP_ONLY=1 python scripts/w4_check.py time 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_ppr_time_old.txt
echo "== new schedule again"; W4_PP_ONLY=1 python scripts/w4_check.py time 2>&1 | grep -v amdgpu.ids | tail -1
DXA_LIB=_abl/lib_stamp.so python scripts/pp_stamps.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_pp_stamps_after.txt
