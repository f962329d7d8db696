#!/bin/bash
# already-built, parity-tested options as defaults?  alternating A/Bs in ONE box:
#   DXA_GEMM_T128_F32EPI=1  the fp32 heads' bf16x3 products (1088 rows) on the 128 x 128-tile kernel          (CogACT, MemVLA)
#   DXA_SWIGLU_FUSE=1       SiLU * up in the gate/up product's epilogue in the TRAINING forward as well     (CogACT)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06_env_ab
O=gpurun_out/r06_env_ab; rm -f $O/*.txt
cg() { env "$@" timeout 300 python bench.py --steps 12 --warmup 3 --no-recipe --no-cpu-baseline --no-secondary --no-latency --no-dp-emulation 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'])"; }
for i in 1 2 3; do
  echo "cogact default        $(cg DXA_X=0)" | tee -a $O/cogact.txt
  echo "cogact T128_F32EPI=1  $(cg DXA_GEMM_T128_F32EPI=1)" | tee -a $O/cogact.txt
  echo "cogact SWIGLU_FUSE=1  $(cg DXA_SWIGLU_FUSE=1)" | tee -a $O/cogact.txt
done
for i in 1 2 3; do
  SKIP_INFER=1 timeout 600 python scripts/memvla_bench.py 3 2>&1 | tail -1 | cut -c40-100 | sed 's/^/memvla default        /' | tee -a $O/memvla.txt
  DXA_GEMM_T128_F32EPI=1 SKIP_INFER=1 timeout 600 python scripts/memvla_bench.py 3 2>&1 | tail -1 | cut -c40-100 | sed 's/^/memvla T128_F32EPI=1  /' | tee -a $O/memvla.txt
done
