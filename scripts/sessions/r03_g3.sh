set -x
python -m pytest tests/test_pi0_gpu.py tests/test_memvla_gpu.py tests/test_parity_gpu.py -m gpu -q -p no:cacheprovider > gpurun_out/r03_t3.log 2>&1
grep -E "passed|failed|Error" gpurun_out/r03_t3.log | tail -8
python scripts/pi0_bench.py 2 16 2>&1 | grep "^{" 
python scripts/memvla_bench.py 2 2>&1 | grep "^{"
