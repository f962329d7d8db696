#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
cat > /tmp/f32b.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from dexbotic_amd import kernels as K
M = 1088
for name, m, n, k in [("qkv", M, 2304, 768), ("proj", M, 768, 768), ("fc1", M, 3072, 768), ("fc2", M, 768, 3072), ("L qkv", 408, 3072, 1024), ("L fc1", 408, 4096, 1024), ("L fc2", 408, 1024, 4096)]:
    a, b = torch.randn(m, k, device="cuda"), torch.randn(n, k, device="cuda")
    out = torch.empty(m, n, device="cuda")
    with K.f32_gemm_mode("bf16x3"):
        for _ in range(3):
            K.mm_nt(a, b, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            K.mm_nt(a, b, out=out)
        e1.record()
        torch.cuda.synchronize()
    print(f"{name:6s} M={m} N={n} K={k}: {e0.elapsed_time(e1) / 30 * 1e3:7.1f} us (product + its two operand splits)")
PY
(echo "== pp3"; python /tmp/f32b.py 2>&1 | grep "us ("; echo "== ring"; DXA_GEMM_PP3=0 python /tmp/f32b.py 2>&1 | grep "us (") | tee gpurun_out/r04_pp3_f32head.txt
