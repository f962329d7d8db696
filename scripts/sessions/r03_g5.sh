python -m pytest tests/test_parity_gpu.py tests/test_small_heads_gpu.py tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -x > gpurun_out/r03_t5.log 2>&1
grep -E "passed|failed|Error|error" gpurun_out/r03_t5.log | tail -8
python scripts/infer_bench.py eager 2>&1 | tail -2
DXA_DIT_SAMPLER=0 python scripts/infer_bench.py eager 2>&1 | tail -1
