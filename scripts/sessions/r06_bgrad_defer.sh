#!/bin/bash
# deferred bias / LayerNorm-affine gradient folds + cached arena views + raw stream handle: MemVLA parity suite, then the step alternating
# with DXA_NO_DEFER_BGRAD=1 (one column sum per consumer) in ONE box, and CogACT
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06_bgrad
O=gpurun_out/r06_bgrad; rm -f $O/*.txt
python - <<'P' 2>&1 | grep -v amdgpu.ids | tee $O/stream_call.txt
import timeit, torch
torch.cuda.set_device(0); torch.zeros(1, device="cuda")
from dexbotic_amd import kernels as K
n = 200000
print("raw handle   %.2f us" % (timeit.timeit(K._stream, number=n) / n * 1e6))
print("Stream object %.2f us" % (timeit.timeit(lambda: torch.cuda.current_stream().cuda_stream, number=n) / n * 1e6))
assert K._stream() == torch.cuda.current_stream().cuda_stream
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    assert K._stream() == s.cuda_stream
print("handles agree on the default and on a side stream")
P
timeout 900 python -m pytest tests/test_memvla_gpu.py tests/test_parity_gpu.py tests/test_zz_dp_gpu.py tests/test_pi0_gpu.py -x -q -m gpu > $O/tests.txt 2>&1; grep -E "passed|failed|error" $O/tests.txt | tail -2
for i in 1 2 3; do
  SKIP_INFER=1 timeout 600 python scripts/memvla_bench.py 3 2>&1 | tail -1 | cut -c1-120 | sed 's/^/deferred     /' | tee -a $O/memvla.txt
  DXA_NO_DEFER_BGRAD=1 SKIP_INFER=1 timeout 600 python scripts/memvla_bench.py 3 2>&1 | tail -1 | cut -c1-120 | sed 's/^/per consumer /' | tee -a $O/memvla.txt
done
for i in 1 2; do
  timeout 300 python bench.py --steps 12 --warmup 3 --no-recipe --no-cpu-baseline --no-secondary --no-latency --no-dp-emulation 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('cogact ms/step', d['ms_per_step'])" | tee -a $O/cogact.txt
done
