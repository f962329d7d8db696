set -x
mkdir -p gpurun_out/prof
python -m pytest tests/test_hf_trainer_gpu.py tests/test_pi0_gpu.py -m gpu -q -p no:cacheprovider > gpurun_out/r03_t2.log 2>&1
tail -15 gpurun_out/r03_t2.log
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o mem -- python $R/scripts/memvla_bench.py 2 > $R/gpurun_out/r03_memvla.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o pi0 -- python $R/scripts/pi0_bench.py 2 16 > $R/gpurun_out/r03_pi0.log 2>&1
cd $R
ls gpurun_out/prof
python profiles/rocpd_stats.py gpurun_out/prof/mem_results.db > gpurun_out/r03_memvla_kernel_stats.txt 2>&1
python profiles/rocpd_stats.py gpurun_out/prof/pi0_results.db > gpurun_out/r03_pi0_kernel_stats.txt 2>&1
rm -rf gpurun_out/prof
tail -3 gpurun_out/r03_memvla.log; tail -3 gpurun_out/r03_pi0.log
head -30 gpurun_out/r03_memvla_kernel_stats.txt
