#!/bin/bash
# round 6: MemVLA fine-tune step in one box: host thread pool left alone (DXA_HOST_THREADS=0) vs inside the cgroup quota; perceptual
# tokens repeated (the reference's layout) vs distinct
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_memvla_dedup; mkdir -p $O; rm -f $O/ab_*.txt
for i in 1 2 3; do
  DXA_HOST_THREADS=0 DXA_MEMVLA_PER_REPEAT=1 SKIP_INFER=1 timeout 600 python scripts/memvla_bench.py 10 2>&1 | tail -1 >> $O/ab_pool_untouched_repeated.txt
  DXA_MEMVLA_PER_REPEAT=1 SKIP_INFER=1 timeout 600 python scripts/memvla_bench.py 10 2>&1 | tail -1 >> $O/ab_repeated.txt
  DXA_MEMVLA_PER_REPEAT=0 SKIP_INFER=1 timeout 600 python scripts/memvla_bench.py 10 2>&1 | tail -1 >> $O/ab_distinct.txt
done
for f in ab_pool_untouched_repeated ab_repeated ab_distinct; do echo $f; cut -c47-110,150-300 $O/$f.txt; done
