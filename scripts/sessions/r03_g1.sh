set -x
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r03_t1.log 2>&1
tail -30 gpurun_out/r03_t1.log
timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-secondary --no-latency > gpurun_out/r03_b_ov.json 2> gpurun_out/r03_b_ov.err
timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-secondary --no-latency --no-overlap --no-recipe > gpurun_out/r03_b_noov.json 2> gpurun_out/r03_b_noov.err
timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-secondary --no-latency --no-overlap --no-recipe --static-batch > gpurun_out/r03_b_static.json 2> gpurun_out/r03_b_static.err
tail -c 600 gpurun_out/r03_b_ov.json; tail -c 300 gpurun_out/r03_b_ov.err
tail -c 400 gpurun_out/r03_b_noov.json; tail -c 400 gpurun_out/r03_b_static.json
