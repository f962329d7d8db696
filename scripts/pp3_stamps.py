#!/usr/bin/env python
"""where a workgroup of the 192-row ping-pong kernel spends its time (tuning build: scripts/build_variant.sh pp3s<N> -DDXA_PP3_STAMPS=<N>,
DXA_LIB=_abl/lib_pp3s<N>.so): cycles entry -> first operands landed -> main loop done -> epilogue stores acknowledged.
Shape from the environment: MNK=543,37888,3584 (default: the request's gate_up)"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dexbotic_amd import kernels as K, _lib as L  # noqa: E402
M, N, Kd = [int(v) for v in os.environ.get("MNK", "543,37888,3584").split(",")]
ws = [(torch.randn(N, Kd, device="cuda") * 0.02).bfloat16() for _ in range(4)]
x = torch.randn(M, Kd, device="cuda").bfloat16()
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
lib = ctypes.CDLL(L.LIB_PATH)
for rep in range(3):
    for w in ws:
        K.mm_nt(x, w, out=out)
    torch.cuda.synchronize()
    st = (ctypes.c_ulonglong * 4)()
    assert lib.dxa_gemm_debug_pp3_stamps(st) == 0
    d = [st[i + 1] - st[i] for i in range(3)]
    print(f"M={M} N={N} K={Kd}: prologue (entry -> first operands landed) {d[0]}  main loop {d[1]} ({d[1] / (Kd // 64):.0f} per K tile of the full K)  "
          f"epilogue {d[2]}  total {st[3] - st[0]} cycles (100 MHz counter: x10 ns)")
