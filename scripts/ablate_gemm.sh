#!/bin/bash
# Kernel-tuning aid: builds libdexbotic_amd variants with parts of the ring GEMM main loop removed
# (DXA_ABL=1 no LDS-DMA, 2 no s_barrier, 3 no ds_read, 4 no MFMA; results are garbage, timings are not)
# into _abl/, to be selected with DXA_LIB=... python scripts/gemm_bench.py
set -e
cd "$(dirname "$0")/.."
python dexbotic_amd/build.py
mkdir -p _abl
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DDXA_ABL=$n -c dexbotic_amd/csrc/gemm.hip -o _abl/gemm_$n.o
  objs=$(ls dexbotic_amd/csrc/_obj/*.o | grep -v gemm.hip.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _abl/lib_$n.so _abl/gemm_$n.o $objs
  echo built _abl/lib_$n.so
done
