#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
python scripts/gemm_hash.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_hash_new.txt
DXA_LIB=_abl/lib_old.so python scripts/gemm_hash.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_hash_old.txt
python scripts/gemm_hash.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_hash_new2.txt
wc -l gpurun_out/r04_hash_new.txt; diff gpurun_out/r04_hash_new.txt gpurun_out/r04_hash_old.txt && echo "NEW == OLD bit for bit"; diff gpurun_out/r04_hash_new.txt gpurun_out/r04_hash_new2.txt && echo "run-to-run identical"
tail -3 gpurun_out/r04_hash_new.txt
