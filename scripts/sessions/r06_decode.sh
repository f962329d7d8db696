#!/bin/bash
# round 6: persistent decode step — parity tests, then the full-size decode bench with the step on / off in one box
out=gpurun_out/r06_decode.txt
: > $out
timeout 900 python -m pytest tests/test_decode_fused_gpu.py tests/test_lm_gpu.py -x -q -s 2>&1 | tail -40 >> $out
echo "# decode_bench, persistent step ON" >> $out
timeout 600 python scripts/decode_bench.py 32 2>&1 | tail -8 >> $out
echo "# decode_bench, persistent step OFF (DXA_DECODE_FUSED=0)" >> $out
DXA_DECODE_FUSED=0 timeout 600 python scripts/decode_bench.py 32 2>&1 | tail -8 >> $out
echo "# json" >> $out
timeout 600 python scripts/decode_bench.py 32 --json 2>&1 | tail -1 >> $out
cat $out
