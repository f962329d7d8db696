#!/bin/bash
# whole GPU suite on the current tree (twice: the capture-time abort was a matter of garbage-collector timing), then pi0 and MemVLA with their new defaults
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06_suite
O=gpurun_out/r06_suite; rm -f $O/gpu_tests*.txt $O/defaults.txt
timeout 1500 python -m pytest tests -q -m gpu > $O/gpu_tests.txt 2>&1; grep -E "passed|failed|error|Fatal" $O/gpu_tests.txt | tail -2
timeout 1500 python -m pytest tests -q -m gpu -p no:randomly > $O/gpu_tests_2.txt 2>&1; grep -E "passed|failed|error|Fatal" $O/gpu_tests_2.txt | tail -2
SKIP_INFER=1 timeout 600 python scripts/pi0_bench.py 3 16 2>&1 | tail -1 | cut -c1-110 | tee -a $O/defaults.txt
SKIP_INFER=1 timeout 600 python scripts/memvla_bench.py 3 2>&1 | tail -1 | cut -c1-110 | tee -a $O/defaults.txt
