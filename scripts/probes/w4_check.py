#!/usr/bin/env python
"""w4 kernel (csrc/gemm_w4.hip) against the ping-pong kernel and an fp32 torch product: correctness on the step's shapes and on
ragged / tail / split cases, then interleaved timing (random bf16 operands in [-1, 1)).

    python scripts/w4_check.py [check] [time] [lib]      # lib: also time torch.matmul (hipBLASLt yardstick, never product)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dexbotic_amd import kernels as K  # noqa: E402

M = int(os.environ.get("W4_M", "4592"))          # W4_M=2296: the reference recipe's micro-batch (8 episodes)
STEP = [("qkv fwd", "nt", M, 4608, 3584), ("o_proj fwd", "nt", M, 3584, 3584), ("gate_up fwd", "nt", M, 37888, 3584),
        ("down fwd", "nt", M, 3584, 18944), ("qkv dX", "nn", M, 3584, 4608), ("o_proj dX", "nn", M, 3584, 3584),
        ("gate_up dX", "nn", M, 3584, 37888), ("down dX", "nn", M, 18944, 3584),
        ("qkv dW", "tn", 4608, 3584, M), ("o_proj dW", "tn", 3584, 3584, M),
        ("gate_up dW", "tn", 37888, 3584, M), ("down dW", "tn", 3584, 18944, M),
        ("sq 4096", "nt", 4096, 4096, 4096), ("sq 8192", "nt", 8192, 8192, 8192)]
EDGE = [("ragged", "nt", 300, 520, 192), ("ragged", "nn", 300, 520, 192), ("ragged", "tn", 300, 520, 200),
        ("one tile", "nt", 256, 256, 64), ("k128", "nn", 513, 264, 128), ("tn tail", "tn", 777, 1032, 4592),
        ("split", "nt", 2296, 3584, 3584), ("split", "nn", 2296, 3584, 18944), ("split", "tn", 4608, 3584, 2296)]


def operands(lay, m, n, k, dev="cuda"):
    g = torch.Generator(device=dev).manual_seed(m * 7 + n * 3 + k)
    r = lambda *s: (torch.rand(*s, device=dev, generator=g) * 2 - 1).bfloat16()
    if lay == "nt":
        return r(m, k), r(n, k), K.mm_nt
    if lay == "nn":
        return r(m, k), r(k, n), K.mm_nn
    return r(k, m), r(k, n), K.mm_tn


def ref32(lay, a, b):
    a, b = a.float(), b.float()
    return a @ b.t() if lay == "nt" else (a @ b if lay == "nn" else a.t() @ b)


def run(fn, a, b, out, w4, **kw):
    os.environ["DXA_GEMM_W4"] = "1" if w4 else "0"
    return fn(a, b, out=out, **kw)


def check():
    bad = 0
    for name, lay, m, n, k in STEP[:12] + EDGE:
        a, b, fn = operands(lay, m, n, k)
        ref = ref32(lay, a, b)
        for odt in ((torch.bfloat16, torch.float32) if lay != "nt" or m < 4000 else (torch.bfloat16,)):
            o_pp = torch.empty(m, n, device="cuda", dtype=odt)
            o_w4 = torch.full((m, n), float("nan"), device="cuda", dtype=odt)
            run(fn, a, b, o_pp, False)
            run(fn, a, b, o_w4, True)
            torch.cuda.synchronize()
            e_ref = ((o_w4.float() - ref).abs().max() / ref.abs().max()).item()
            same = torch.equal(o_pp, o_w4)
            e_pp = ((o_w4.float() - o_pp.float()).abs().max() / ref.abs().max()).item()
            ok = e_ref < (1e-2 if odt == torch.bfloat16 else 2e-5 * max(1, k // 1024)) and not torch.isnan(o_w4).any().item()
            bad += not ok
            print(f"{'ok ' if ok else 'BAD'} {name:12s} {lay} {m:6d} {n:6d} {k:6d} {str(odt)[6:]:9s} vs fp32 {e_ref:.2e}  vs pp {e_pp:.2e} "
                  f"{'bit-identical' if same else ''}", flush=True)
            if not ok:
                d = (o_w4.float() - ref).abs()
                idx = torch.nonzero(d > 0.05 * ref.abs().max())[:6].tolist()
                print("     first mismatches (row, col):", idx, " nan:", int(torch.isnan(o_w4).sum()), flush=True)
                rows = torch.nonzero((d > 0.05 * ref.abs().max()).any(1)).flatten()
                cols = torch.nonzero((d > 0.05 * ref.abs().max()).any(0)).flatten()
                print("     bad rows:", rows[:12].tolist(), "...", int(rows.numel()), " bad cols:", cols[:12].tolist(), "...", int(cols.numel()), flush=True)
    # epilogue menu on the w4 path: bias + residual, accumulate, fp32 C with mirror + sumsq
    m, n, k = 1000, 776, 512
    a, b, fn = operands("nt", m, n, k)
    bias = (torch.rand(n, device="cuda") - 0.5).bfloat16()
    res = (torch.rand(m, n, device="cuda") - 0.5).bfloat16()
    want = ref32("nt", a, b) * 0.5 + bias.float() + res.float()
    o = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    run(fn, a, b, o, True, bias=bias, residual=res, alpha=0.5)
    e = ((o.float() - want).abs().max() / want.abs().max()).item()
    print(f"{'ok ' if e < 1e-2 else 'BAD'} epilogue bias+residual+alpha {e:.2e}")
    bad += e >= 1e-2
    a, b, fn = operands("tn", 520, 776, 1000)
    c0 = torch.rand(520, 776, device="cuda")
    o = c0.clone()
    mir = torch.empty(520, 776, device="cuda", dtype=torch.bfloat16)
    ssq = torch.zeros(K.L.lib.dxa_gemm_sumsq_slots(520, 776), device="cuda")
    run(fn, a, b, o, True, accumulate=True, mirror=mir, sumsq=ssq)
    want = ref32("tn", a, b) + c0
    e = ((o - want).abs().max() / want.abs().max()).item()
    e2 = ((mir.float() - want).abs().max() / want.abs().max()).item()
    e3 = abs(ssq.sum().item() - (o.double() ** 2).sum().item()) / (o.double() ** 2).sum().item()
    print(f"{'ok ' if max(e, e3) < 1e-4 and e2 < 1e-2 else 'BAD'} epilogue accumulate {e:.2e} mirror {e2:.2e} sumsq {e3:.2e}")
    bad += not (max(e, e3) < 1e-4 and e2 < 1e-2)
    print("CHECK", "FAILED" if bad else "PASSED", bad)
    return bad


def timeit(f, reps=10):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def bench(lib):
    tot = {"pp": 0.0, "w4": 0.0, "lib": 0.0}
    for name, lay, m, n, k in STEP:
        a, b, fn = operands(lay, m, n, k)
        out = torch.empty(m, n, device="cuda", dtype=torch.float32 if lay == "tn" else torch.bfloat16)
        res = {}
        for rnd in range(2):                 # interleaved rounds, best of two
            for tag in (("pp",) if os.environ.get("W4_PP_ONLY") else ("pp", "w4")) + (("lib",) if lib else ()):
                if tag == "lib":
                    ta, tb = (a.t() if lay == "tn" else a), (b.t() if lay == "nt" else b)
                    lout = out if out.dtype == torch.bfloat16 else torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
                    us = timeit(lambda: torch.matmul(ta, tb, out=lout))
                else:
                    us = timeit(lambda: run(fn, a, b, out, tag == "w4"))
                res[tag] = min(res.get(tag, 1e30), us)
        fl = 2 * m * n * k
        line = f"{name:12s} {lay} M={m:6d} N={n:6d} K={k:6d}"
        for tag, us in res.items():
            line += f" | {tag} {us:8.1f} us {fl / us / 1e6:7.1f} TF/s"
            if not name.startswith("sq"):
                tot[tag] += us
        print(line, flush=True)
    print("layer total (12 products): " + "  ".join(f"{t} {v / 1e3:.3f} ms" for t, v in tot.items() if v), flush=True)


def ablate():
    """NT bf16 tuning instantiations of the w4 kernel (DXA_GEMM_W4V): what each part of the K loop costs"""
    names = {0: "schedule 0", 32: "no barrier", 64: "no vmcnt wait", 100: "schedule 1", 132: "schedule 1 no barrier"}
    for name, lay, m, n, k in [STEP[1], STEP[2], STEP[3], STEP[13]]:
        a, b, fn = operands(lay, m, n, k)
        out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
        res = {}
        for rnd in range(2):
            for v in names:
                os.environ["DXA_GEMM_W4V"] = str(v)
                res[v] = min(res.get(v, 1e30), timeit(lambda: run(fn, a, b, out, True)))
            os.environ["DXA_GEMM_W4V"] = "0"
            res["pp"] = min(res.get("pp", 1e30), timeit(lambda: run(fn, a, b, out, False)))
        print(f"{name:12s} {lay} M={m} N={n} K={k}: pp {res['pp']:.1f} us | " + " | ".join(f"{names[v]} {res[v]:.1f}" for v in names), flush=True)


def pmc_run():
    """a few launches of pp / w4 / library on two shapes, for a rocprofv3 --pmc pass (scripts/r04_g3.sh)"""
    for name, lay, m, n, k in [STEP[3], STEP[2]]:
        a, b, fn = operands(lay, m, n, k)
        out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
        for _ in range(4):
            run(fn, a, b, out, False)
            run(fn, a, b, out, True)
            torch.matmul(a, b.t(), out=out)
        torch.cuda.synchronize()


if __name__ == "__main__":
    what = sys.argv[1:] or ["check", "time"]
    rc = 0
    for w in what:
        if w.startswith("check") and w != "check":      # check200: the NT bf16 tuning instantiation DXA_GEMM_W4V=200 etc.
            os.environ["DXA_GEMM_W4V"] = w[5:]
            rc |= check()
            os.environ["DXA_GEMM_W4V"] = "0"
    if "check" in what:
        rc |= check()
    if "time" in what:
        bench("lib" in what)
    if "abl" in what:
        ablate()
    if "pmc" in what:
        pmc_run()
    sys.exit(1 if rc else 0)
