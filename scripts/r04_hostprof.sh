#!/bin/bash
# host-side profile of the recipe step (8 episodes x 2 accumulation steps): where the Python launch path spends its time
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
python -c "
import cProfile, pstats, sys, io
sys.argv = ['bench.py', '--batch', '16', '--accum', '2', '--steps', '8', '--warmup', '2', '--no-cpu-baseline', '--no-latency', '--no-secondary', '--no-recipe']
import runpy
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path('bench.py', run_name='__main__')
except SystemExit:
    pass
pr.disable()
s = io.StringIO()
ps = pstats.Stats(pr, stream=s).sort_stats('tottime')
ps.print_stats(45)
print(s.getvalue()[:9000])
s = io.StringIO()
ps = pstats.Stats(pr, stream=s).sort_stats('cumulative')
ps.print_stats(60)
print(s.getvalue()[:12000])
" > gpurun_out/r04_hostprof.txt 2>&1
head -75 gpurun_out/r04_hostprof.txt | cut -c1-180
