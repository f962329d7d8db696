echo "== default"; timeout 100 python scripts/gemm_bench.py "pre" 2>&1 | grep -v amdgpu
echo "== AI4 (pp kernel, 256-row tiles)"; DXA_GEMM_RING_AI=4 timeout 100 python scripts/gemm_bench.py "pre" 2>&1 | grep -v amdgpu
echo "== AI4 split max 16, min piece 8"; DXA_GEMM_RING_AI=4 DXA_SPLIT_MAX=16 DXA_SPLIT_MIN_PIECE=8 timeout 100 python scripts/gemm_bench.py "pre" 2>&1 | grep -v amdgpu
echo "== AI3 split max 16, min piece 8"; DXA_SPLIT_MAX=16 DXA_SPLIT_MIN_PIECE=8 timeout 100 python scripts/gemm_bench.py "pre" 2>&1 | grep -v amdgpu
