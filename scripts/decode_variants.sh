#!/bin/bash
# Round 5: the decode step's weight-streaming kernel (gemm_skinny_bf16_kernel, 72 % of a token): 4 blocks of 64 k in flight per
# wave (hand-unrolled; the default) against the one block of rounds 2-4 (DXA_SKINNY_BF16_UNROLL=1), one box.
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
out=gpurun_out/r05_decode_unroll.txt; : > $out
for e in "DXA_SKINNY_BF16_UNROLL=1" "DXA_NONE=0" "DXA_SKINNY_BF16_UNROLL=1" "DXA_NONE=1"; do
  echo "== decode_bench $e" >> $out
  env $e timeout 300 python scripts/decode_bench.py 32 --json 2>&1 | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); g=d['greedy_ids_vs_uncached_reforward']
print(d['ms_per_token'], d['ms_per_token_all'], 'prefill+1', d['prefill_plus_first_token_ms'], 'ids agree', g['agree_prefix'], '/', g['checked'], g['cached'])" >> $out
done
cat $out
