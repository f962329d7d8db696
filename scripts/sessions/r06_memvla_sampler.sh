#!/bin/bash
# round 6: MemVLA's DiT-L sampler (perceptual attention) as one persistent launch: kernel + model tests, the sampler alone (with / without the
# perceptual-attention phases; DXA_DIT_DBG=3: without the attention work, =7: keys / values requested after the barrier), per-frame latency A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_memvla_sampler; mkdir -p $O; rm -f $O/*.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -s -k "dit_sample_bf16 or dit_bf16_pack" 2>&1 | grep "sampler\|passed\|failed" | tee $O/kernel_tests.txt
timeout 1500 python -m pytest tests/test_memvla_gpu.py -q -x -s -k "one_launch" 2>&1 | grep "raw\|frame\|passed\|failed" | tee $O/model_tests.txt
for d in 0 3 7; do DXA_DIT_DBG=$d timeout 300 python scripts/probes/memvla_sampler_time.py 2>&1 | grep "DiT-L" | tee -a $O/sampler_alone.txt; done
for i in 1 2; do
  DXA_DIT_SAMPLER=0 SKIP_TRAIN=1 timeout 600 python scripts/memvla_bench.py 2>&1 | tail -1 | cut -c1-300 | tee -a $O/frame_per_block.txt
  SKIP_TRAIN=1 timeout 600 python scripts/memvla_bench.py 2>&1 | tail -1 | cut -c1-300 | tee -a $O/frame_one_launch.txt
done
