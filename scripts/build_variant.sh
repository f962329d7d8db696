#!/bin/bash
# scripts/build_variant.sh NAME [-DFLAG ...]: a tuning build of the library with ONE source (SRC=gemm.hip by default) compiled under
# extra flags -> _abl/lib_NAME.so (git-ignored, travels with gpurun).  The other objects come from the normal in-tree build.
# Use with DXA_LIB=_abl/lib_NAME.so
set -e
cd "$(dirname "$0")/.."
name=$1; shift
src=${SRC:-gemm.hip}
mkdir -p _abl
python -m dexbotic_amd.build > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c dexbotic_amd/csrc/$src -o _abl/v_$name.o
objs=$(ls dexbotic_amd/csrc/_obj/*.o | grep -v "/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _abl/lib_$name.so _abl/v_$name.o $objs
echo "_abl/lib_$name.so"
