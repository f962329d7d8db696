#!/bin/bash
# Kernel-tuning aid: builds libdexbotic_amd variants of the GEMM fast paths into _abl/ (git-ignored, travels with gpurun),
# to be selected with DXA_LIB=_abl/lib_<tag>.so python scripts/gemm_bench.py
#   scripts/ablate_gemm.sh tag1:"-DDXA_PPV=1" tag2:"-DDXA_PP_PH=2" ...
# Ring kernel: -DDXA_ABL=1 no LDS-DMA, 2 no s_barrier, 3 no ds_read, 4 no MFMA.  Ping-pong kernel: -DDXA_PPV bits
# 1 no LDS-DMA, 2 no ds_read, 4 no MFMA, 8 no s_setprio, 16 no stagger; -DDXA_PP_PH=2 two phases per K tile.
# (ablation results are garbage, timings are not)
set -e
cd "$(dirname "$0")/.."
mkdir -p _abl
objs=$(ls dexbotic_amd/csrc/_obj/*.o | grep -v gemm.hip.o)
for spec in "$@"; do
  tag=${spec%%:*}; flags=${spec#*:}
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c dexbotic_amd/csrc/gemm.hip -o _abl/gemm_$tag.o \
    && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _abl/lib_$tag.so _abl/gemm_$tag.o $objs \
    && echo built _abl/lib_$tag.so ) &
done
wait
