#!/bin/bash
# round 5, GPU session 2: 2-rank model step, trajectory / depth-12 pins, clock readout, bench secondary legs
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_zz_dp2_gpu.py tests/test_realwidth_gpu.py -q -s -x > gpurun_out/r05_g2_tests.txt 2>&1
grep -v "^$\|Warning\|warn\|amdgpu.ids\|socket.cpp\|Gloo" gpurun_out/r05_g2_tests.txt | tail -70 | cut -c1-250
# what the box lets a user see / set about clocks and power (read-only first)
(rocm-smi --showperflevel --showclocks --showpower --showmaxpower --showtemp 2>&1 | head -60) > gpurun_out/r05_smi.txt
cat gpurun_out/r05_smi.txt | head -40
python scripts/gemm_bench.py "down    fwd,down    dW ,down    dX " 2>&1 | grep -v amdgpu > gpurun_out/r05_gemm_auto.txt; cat gpurun_out/r05_gemm_auto.txt
(rocm-smi --setperflevel high 2>&1 | tail -3) >> gpurun_out/r05_smi.txt
python scripts/gemm_bench.py "down    fwd,down    dW ,down    dX " 2>&1 | grep -v amdgpu > gpurun_out/r05_gemm_high.txt; cat gpurun_out/r05_gemm_high.txt
(rocm-smi --showperflevel --showclocks 2>&1 | head -30; rocm-smi --setperflevel auto 2>&1 | tail -2) >> gpurun_out/r05_smi.txt
tail -12 gpurun_out/r05_smi.txt
timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-recipe > gpurun_out/r05_g2_bench.json 2> gpurun_out/r05_g2_bench.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r05_g2_bench.json") if l.startswith("{")][-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "p50_action_inference_ms", "action_inference")})
print(json.dumps(d.get("secondary"))[:1500])
PY
