#!/bin/bash
# round 6: the request's few-row products, per shape: default dispatch vs 192-row tiles with every tile cut along K; stamps of one
# producer / one gatherer workgroup of the split qkv product
out=gpurun_out/r06_m543.txt
: > $out
echo "# default dispatch" >> $out
ROWS=543 python scripts/prefill_gemm_bench.py >> $out 2>&1
echo "# DXA_GEMM_NO_T128=1 (192-row tiles, all tiles cut along K)" >> $out
DXA_GEMM_NO_T128=1 ROWS=543 python scripts/prefill_gemm_bench.py 2>&1 | head -2 >> $out
echo "# DXA_GEMM_NO_T128=1 DXA_SPLIT_MAX=2" >> $out
DXA_GEMM_NO_T128=1 DXA_SPLIT_MAX=2 ROWS=543 python scripts/prefill_gemm_bench.py 2>&1 | head -2 >> $out
echo "# DXA_GEMM_NO_T128=1 DXA_GEMM_NO_SPLIT=1" >> $out
DXA_GEMM_NO_T128=1 DXA_GEMM_NO_SPLIT=1 ROWS=543 python scripts/prefill_gemm_bench.py 2>&1 | head -2 >> $out
echo "# ViT shapes, 514 rows, default" >> $out
SHAPES=vit ROWS=514 python scripts/prefill_gemm_bench.py >> $out 2>&1
for b in 0 200; do
  echo "# stamps of workgroup $b (qkv, NO_T128: 54 tiles x 4 K pieces; block >= 162 gathers)" >> $out
  DXA_LIB=_abl/lib_pp3s$b.so DXA_GEMM_NO_T128=1 MNK=543,4608,3584 python scripts/pp3_stamps.py >> $out 2>&1
done
cat $out
