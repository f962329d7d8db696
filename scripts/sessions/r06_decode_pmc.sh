#!/bin/bash
# HBM-side bytes of the persistent decode step per launch (FETCH_SIZE x2 on gfx950, WRITE_SIZE), own --pmc passes
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/decode_pmc; mkdir -p $O
out=gpurun_out/r06_decode_pmc.txt
: > $out
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $O -o d_$c -- python scripts/decode_bench.py 9 > $O/d_$c.log 2>&1
  echo "# $c (unit 1024 B as reported; FETCH_SIZE x2 on gfx950)" >> $out
  python profiles/rocpd_stats.py --pmc $O/d_${c}_results.db "decode_step_k,gemm_skinny_bf16_kernel" >> $out 2>&1
done
rm -f $O/*.db
grep -v "counters_collection columns\|^W2026" $out | cut -c1-200
