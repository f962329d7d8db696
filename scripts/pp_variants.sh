#!/bin/bash
# runs scripts/gemm_bench.py on a few shapes for every _abl/lib_*.so (GPU box)
for lib in _abl/lib_*.so; do
  echo "== $lib"
  DXA_LIB=$lib timeout 120 python scripts/gemm_bench.py "$@" 2>&1 | grep -v amdgpu.ids
done
