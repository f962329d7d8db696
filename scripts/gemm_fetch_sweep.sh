#!/bin/bash
# L2-miss (fabric-side) bytes of the ping-pong GEMM per tile-order group size, against the arithmetic of DESIGN.md section 4
# ("what the 3.6x is"): FETCH_SIZE / WRITE_SIZE in their own --pmc passes over scripts/gemm_bench.py on the gate_up forward (NT), the
# gate_up dX (NN) and the gate_up dW (TN) product, DXA_GEMM_GROUP_M = 2, 4 (default), 8, 18.  -> gpurun_out/r06_fetch_by_group.txt
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/fetch_sweep; mkdir -p $O
out=gpurun_out/r06_fetch_by_group.txt
: > $out
for g in 2 4 8 18; do
  for c in FETCH_SIZE WRITE_SIZE; do
    DXA_GEMM_GROUP_M=$g rocprofv3 --pmc $c --kernel-trace -d $O -o g${g}_$c -- python scripts/gemm_bench.py "gate_up fwd,gate_up dX,gate_up dW" > $O/g${g}_$c.log 2>&1
    echo "# DXA_GEMM_GROUP_M=$g $c (unit 1024 B as reported; FETCH_SIZE x2 on gfx950)" >> $out
    python profiles/rocpd_stats.py --pmc $O/g${g}_${c}_results.db "gemm_pp_kernel" >> $out 2>&1
  done
  grep "gate_up" $O/g${g}_FETCH_SIZE.log | head -3 >> $out
done
rm -f $O/*.db
cat $out | cut -c1-200
