set -x
mkdir -p gpurun_out/prof
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for n in 1 3; do
SKIP_INFER=1 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o mem$n -- python $R/scripts/memvla_bench.py $n > $R/gpurun_out/r03_memvla_$n.log 2>&1
SKIP_INFER=1 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o pi$n -- python $R/scripts/pi0_bench.py $n 16 > $R/gpurun_out/r03_pi0_$n.log 2>&1
done
cd $R
python profiles/rocpd_stats.py --per-step gpurun_out/prof/mem1_results.db 3 gpurun_out/prof/mem3_results.db 5 > gpurun_out/r03_memvla_per_step.txt 2>&1
python profiles/rocpd_stats.py --per-step gpurun_out/prof/pi1_results.db 3 gpurun_out/prof/pi3_results.db 5 > gpurun_out/r03_pi0_per_step.txt 2>&1
rm -rf gpurun_out/prof
grep "^{" gpurun_out/r03_memvla_3.log gpurun_out/r03_pi0_3.log
head -45 gpurun_out/r03_memvla_per_step.txt | cut -c1-170
