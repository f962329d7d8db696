#!/bin/bash
# Round 5: A/B of the attention kernel variants inside ONE box (boxes differ by several per cent): micro-benchmark with output
# fingerprints (variants of one kernel accumulate in the same order: the fingerprints must be equal), the attention unit tests
# under every variant, the pi0 fine-tune step under the candidates.   gpurun -- 'bash scripts/attn_variants.sh'
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
out=gpurun_out/r05_attn_variants.txt; : > $out
export ATTN_BENCH_SUMS=1
for e in "DXA_NONE=0" "DXA_ATTN_FWD_TR=1" "DXA_ATTN_DKV256=1" "DXA_ATTN_DKV256=2" "DXA_ATTN_DKV256=3" "DXA_ATTN_DKV128=2"; do
  echo "== attn_bench $e" >> $out
  env $e timeout 300 python scripts/attn_bench.py 2>&1 | grep -v "^ROCm version\|^Hostname\|^Librccl\|^RCCL\|^HIP version" >> $out
done
for e in "DXA_ATTN_FWD_TR=1" "DXA_ATTN_DKV256=1" "DXA_ATTN_DKV256=2" "DXA_ATTN_DKV256=3" "DXA_ATTN_DKV128=2"; do
  echo "== tests $e" >> $out
  env $e timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "attention" 2>&1 | tail -2 >> $out
done
for e in "DXA_NONE=0" "DXA_ATTN_FWD_TR=1" "DXA_ATTN_FWD_TR=1 DXA_ATTN_DKV256=2" "DXA_ATTN_FWD_TR=1 DXA_ATTN_DKV256=3" "DXA_ATTN_FWD_TR=1 DXA_ATTN_DKV256=1"; do
  echo "== pi0 step $e" >> $out
  env $e SKIP_INFER=1 timeout 300 python scripts/pi0_bench.py 6 16 2>&1 | grep "^{" >> $out
done
cat $out
