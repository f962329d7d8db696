for g in 1 2 4 8 16; do echo "== group_m $g"; DXA_GEMM_GROUP_M=$g timeout 100 python scripts/gemm_bench.py "gate_up fwd,down    fwd,gate_up dW,gate_up dX,8192" 2>&1 | grep -v amdgpu; done
echo "== dmafirst"; DXA_LIB=_abl/lib_dmafirst.so timeout 100 python scripts/gemm_bench.py "gate_up fwd,down    fwd,gate_up dW,gate_up dX,8192" 2>&1 | grep -v amdgpu
