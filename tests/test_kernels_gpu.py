"""Kernel-level numerics: every C-ABI entry point against a plain torch fp32/fp64 reference of the
same op, on the MI355X.  (Model-level parity against the oracle / golden vectors: test_parity_gpu.py.)"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from dexbotic_amd import _lib as L
    from dexbotic_amd import kernels as K

DEV = "cuda"


def rnd(*shape, dtype=torch.float32, scale=1.0, seed=None):
    g = torch.Generator(device="cpu")
    g.manual_seed(seed if seed is not None else (hash(shape) & 0xFFFF))
    return (torch.randn(*shape, generator=g) * scale).to(DEV).to(dtype)


def assert_close(out, ref, rtol, atol, what=""):
    out64, ref64 = out.double(), ref.double()
    err = (out64 - ref64).abs()
    tol = atol + rtol * ref64.abs()
    bad = err > tol
    if bad.any():
        i = torch.nonzero(bad)[0].tolist()
        raise AssertionError(f"{what}: {int(bad.sum())}/{bad.numel()} mismatches, max err {err.max().item():.3e} "
                             f"(ref max {ref64.abs().max().item():.3e}); first at {i}: got {out64[tuple(i)].item():.6e} "
                             f"want {ref64[tuple(i)].item():.6e}")


def tol_for(dtype, K_=1):
    if dtype == torch.bfloat16:
        return 1.0 / 128, 1e-2 * math.sqrt(max(K_, 1)) / 16
    return 2e-5, 6e-6 * math.sqrt(max(K_, 1))


ACTS = {6: lambda x: torch.sigmoid(x), 0: lambda x: x, 1: lambda x: F.gelu(x), 2: lambda x: F.gelu(x, approximate="tanh"),
        3: lambda x: x * torch.sigmoid(1.702 * x), 4: F.silu, 5: F.relu}


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("layout", ["nt", "nn", "tn"])
@pytest.mark.parametrize("shape", [(128, 128, 64), (256, 384, 256), (200, 136, 96), (130, 70, 588), (17, 7, 7),
                                   (1000, 520, 1024), (64, 2304, 768)])
def test_gemm_plain(dtype, layout, shape):
    M, N, Kd = shape
    a = rnd(M, Kd, dtype=dtype, seed=1) if layout != "tn" else rnd(Kd, M, dtype=dtype, seed=1)
    b = rnd(N, Kd, dtype=dtype, seed=2) if layout == "nt" else rnd(Kd, N, dtype=dtype, seed=2)
    fn = {"nt": K.mm_nt, "nn": K.mm_nn, "tn": K.mm_tn}[layout]
    out = fn(a, b)
    A = a.double() if layout != "tn" else a.double().t()
    Bm = b.double().t() if layout == "nt" else b.double()
    ref = A @ Bm
    rtol, atol = tol_for(dtype, Kd)
    assert out.shape == (M, N) and out.dtype == dtype
    assert_close(out, ref, rtol, atol, f"gemm {layout} {shape} {dtype}")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("act", [0, 1, 2, 3, 4])
def test_gemm_epilogue(dtype, act):
    M, N, Kd = 300, 264, 192
    a, w = rnd(M, Kd, dtype=dtype, seed=3), rnd(N, Kd, dtype=dtype, seed=4, scale=0.1)
    bias, res = rnd(N, dtype=dtype, seed=5), rnd(M, N, dtype=dtype, seed=6)
    aux = torch.empty(M, N, device=DEV, dtype=dtype)
    out = K.mm_nt(a, w, bias=bias, act=act, residual=res, aux_out=aux)
    pre = a.double() @ w.double().t() + bias.double()
    ref = ACTS[act](pre) + res.double()
    rtol, atol = tol_for(dtype, Kd)
    assert_close(aux, pre, rtol, atol, "aux_out")
    assert_close(out, ref, 2 * rtol, 2 * atol, f"epilogue act={act}")
    # mulgrad: out = (dy @ w) * act'(pre)
    dy = rnd(M, N, dtype=dtype, seed=7)
    pre_t = pre.to(dtype)
    g = K.mm_nn(dy, w, mulgrad=rnd(M, Kd, dtype=dtype, seed=8), act=act)
    xg = rnd(M, Kd, dtype=dtype, seed=8).double().requires_grad_(True)
    ACTS[act](xg).backward(dy.double() @ w.double())
    assert_close(g, xg.grad, 3 * rtol, 3 * atol, f"mulgrad act={act}")
    del pre_t


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_accumulate_f32out_strided(dtype):
    M, N, Kd = 260, 200, 320
    big_a = rnd(M, Kd + 24, dtype=dtype, seed=9)
    a = big_a[:, 8:8 + Kd] if dtype == torch.float32 else big_a[:, 8:8 + Kd]   # row stride != K, 16B-aligned offset
    x = rnd(M, N, dtype=dtype, seed=10)
    out = rnd(Kd, N, dtype=torch.float32, seed=11)
    ref = out.double() + a.double().t() @ x.double()
    K.mm_tn(a, x, out=out, accumulate=True)                     # dW-style: fp32 accumulate into existing grad
    rtol, atol = (2e-5, 2e-5) if dtype == torch.float32 else (2e-5, 2e-3)
    assert_close(out, ref, rtol, atol, "accumulate fp32 out")
    # unaligned view (offset of 1 element): scalar path
    a1 = big_a[:, 1:1 + Kd]
    o2 = K.mm_nt(a1, rnd(N, Kd, dtype=dtype, seed=12))
    ref2 = a1.double() @ rnd(N, Kd, dtype=dtype, seed=12).double().t()
    r2, a2 = tol_for(dtype, Kd)
    assert_close(o2, ref2, r2, a2, "unaligned rows")


@pytest.mark.parametrize("shape", [(1100, 530, 2048), (700, 300, 4096), (4592, 4608, 2048), (4112, 1024, 4096)])
@pytest.mark.parametrize("f32out", [False, True])
def test_gemm_ring_split_tail(shape, f32out):
    """bf16 NT ring kernel when the last round of 256x256 tiles is cut along K (fp32 partials + gatherer)"""
    M, N, Kd = shape
    a, w = rnd(M, Kd, dtype=torch.bfloat16, seed=15), rnd(N, Kd, dtype=torch.bfloat16, seed=16, scale=0.1)
    ref = a.double() @ w.double().t()
    for rep in range(3):                                   # the arrival counters must reset themselves
        if f32out:
            out = torch.full((M, N), 0.5, device=DEV, dtype=torch.float32)
            K.mm_nt(a, w, out=out, accumulate=True)
            assert_close(out, ref + 0.5, 2e-5, 2e-3 * math.sqrt(Kd / 320), f"split tail f32 accumulate rep {rep}")
        else:
            bias, res = rnd(N, dtype=torch.bfloat16, seed=17), rnd(M, N, dtype=torch.bfloat16, seed=18)
            out = K.mm_nt(a, w, bias=bias, residual=res)
            rtol, atol = tol_for(torch.bfloat16, Kd)
            assert_close(out, ref + bias.double() + res.double(), 2 * rtol, 2 * atol, f"split tail bf16 rep {rep}")


@pytest.mark.parametrize("shape", [(1280, 256, 64), (1536, 520, 128), (2024, 520, 192), (1794, 129, 320), (4592, 4608, 3584),
                                   (2304, 1024, 4608), (1536, 2048, 18944), (1724, 300, 4096)])
@pytest.mark.parametrize("f32out", [False, True])
def test_gemm_pingpong(shape, f32out):
    """bf16 NT ping-pong kernel (256-row tiles, K % 64 == 0): 1, 2, 3 and many K tiles (odd and even counts: the two LDS
    buffers alternate), ragged M / N edges, the split-K tail, fused epilogue, fp32 accumulate output; every output element
    against an fp64 reference, several repetitions (the staggered wave groups must never read a K tile early)"""
    M, N, Kd = shape
    # dispatch: ping-pong main loop (more than 1024 rows: below that the few-row 128x128 kernel takes NT products)
    assert M > 1024 and Kd % 64 == 0 and not (-(-M // 192) * 192 * 27 < -(-M // 256) * 256 * 25)
    a, w = rnd(M, Kd, dtype=torch.bfloat16, seed=35), rnd(N, Kd, dtype=torch.bfloat16, seed=36, scale=0.1)
    ref = a.double() @ w.double().t()
    first = None
    for rep in range(3):
        if f32out:
            out = torch.full((M, N), 0.5, device=DEV, dtype=torch.float32)
            K.mm_nt(a, w, out=out, accumulate=True)
            assert_close(out, ref + 0.5, 2e-5, 2e-3 * math.sqrt(Kd / 320), f"ping-pong f32 accumulate rep {rep}")
        else:
            bias, res = rnd(N, dtype=torch.bfloat16, seed=37), rnd(M, N, dtype=torch.bfloat16, seed=38)
            out = K.mm_nt(a, w, bias=bias, residual=res)
            rtol, atol = tol_for(torch.bfloat16, Kd)
            assert_close(out, ref + bias.double() + res.double(), 2 * rtol, 2 * atol, f"ping-pong bf16 rep {rep}")
        if first is None:
            first = out.clone()
        assert torch.equal(out, first), f"ping-pong result changed between repetitions (rep {rep})"


@pytest.mark.parametrize("case", [("nn", 4592, 3584, 4608), ("nn", 1000, 520, 1024), ("nn", 512, 2048, 192), ("nn", 300, 264, 64),
                                  ("tn", 4608, 3584, 4592), ("tn", 1024, 768, 4592), ("tn", 520, 264, 1000), ("tn", 256, 256, 24),
                                  ("tn", 300, 136, 70), ("tn", 2304, 1024, 4112)])
def test_gemm_pingpong_strided_operands(case):
    """bf16 NN (dX = dY W) and TN (dW = dY^T X) on the ping-pong kernel with K-STRIDED operands staged as they lie (LDS-DMA of
    [64 k][128 rows] pieces, ds_read_b64_tr_b16 fragments): every output element against fp64; ragged M / N, K that is not a
    multiple of 64 (TN: rows past K must read as zeros), split-K tails, bf16 output with bias + residual and fp32
    accumulating output (the gradient-arena form), repetitions bitwise equal"""
    lay, M, N, Kd = case
    if lay == "nn":
        a, b = rnd(M, Kd, dtype=torch.bfloat16, seed=45), rnd(Kd, N, dtype=torch.bfloat16, seed=46, scale=0.1)
        ref, fn = a.double() @ b.double(), K.mm_nn
    else:
        a, b = rnd(Kd, M, dtype=torch.bfloat16, seed=45), rnd(Kd, N, dtype=torch.bfloat16, seed=46, scale=0.1)
        ref, fn = a.double().t() @ b.double(), K.mm_tn
    first = {}
    for rep in range(2):
        out32 = torch.full((M, N), 0.25, device=DEV, dtype=torch.float32)
        mirror = torch.zeros((M, N), device=DEV, dtype=torch.bfloat16)
        fn(a, b, out=out32, accumulate=True, mirror=mirror)
        assert_close(out32, ref + 0.25, 2e-5, 2e-3 * math.sqrt(max(Kd, 320) / 320), f"{lay} f32 accumulate {case} rep {rep}")
        # the bf16 communication copy written by the same epilogue (or, off the lean path, by one copy pass)
        assert torch.equal(mirror, out32.to(torch.bfloat16)), f"{lay} mirror {case}"
        bias, res = rnd(N, dtype=torch.bfloat16, seed=47), rnd(M, N, dtype=torch.bfloat16, seed=48)
        out16 = fn(a, b, bias=bias, residual=res)
        rtol, atol = tol_for(torch.bfloat16, Kd)
        assert_close(out16, ref + bias.double() + res.double(), 2 * rtol, 2 * atol, f"{lay} bf16 {case} rep {rep}")
        for k_, o_ in (("f32", out32), ("bf16", out16)):
            if k_ in first:
                assert torch.equal(first[k_], o_), f"{lay} {k_} result changed between repetitions"
            first[k_] = o_.clone()
    if lay == "nn":                                   # the activation-gradient epilogue of the ViT / projector dX products
        pre = rnd(M, N, dtype=torch.bfloat16, seed=49)
        g = K.mm_nn(a, b, mulgrad=pre, act=3)
        xg = pre.double().requires_grad_(True)
        ACTS[3](xg).backward(ref)
        rtol, atol = tol_for(torch.bfloat16, Kd)
        assert_close(g, xg.grad, 3 * rtol, 3 * atol, f"nn mulgrad {case}")


@pytest.mark.parametrize("shape", [(543, 4608, 3584), (543, 3584, 18944), (514, 1024, 4096), (130, 300, 64), (330, 260, 96),
                                   (192, 256, 32), (1050, 777, 2048)])
@pytest.mark.parametrize("f32out", [False, True])
def test_gemm_ring_192_row_tiles(shape, f32out):
    """bf16 NT products on 192-row tiles (chosen when they trim the row padding: the B=1 prefill shapes): the ping-pong 192-row
    kernel (round 4: gemm_pp3_kernel, K % 64 == 0 — lean epilogue, or the generic one where N is ragged) and the ring kernel
    (K % 64 != 0); with and without the split-K tail, ragged M and N edges.  scripts/pp3_check.py: the two kernels agree bit for bit
    wherever they cut K at the same places (profiles/r04_pp3_vs_ring.txt)"""
    M, N, Kd = shape
    assert -(-M // 192) * 192 * 27 < -(-M // 256) * 256 * 25          # dispatch takes the 192-row variant
    a, w = rnd(M, Kd, dtype=torch.bfloat16, seed=25), rnd(N, Kd, dtype=torch.bfloat16, seed=26, scale=0.1)
    ref = a.double() @ w.double().t()
    for rep in range(2):
        if f32out:
            out = torch.full((M, N), 0.5, device=DEV, dtype=torch.float32)
            K.mm_nt(a, w, out=out, accumulate=True)
            assert_close(out, ref + 0.5, 2e-5, 2e-3 * math.sqrt(Kd / 320), f"192-row f32 accumulate rep {rep}")
        else:
            bias, res = rnd(N, dtype=torch.bfloat16, seed=27), rnd(M, N, dtype=torch.bfloat16, seed=28)
            out = K.mm_nt(a, w, bias=bias, residual=res)
            rtol, atol = tol_for(torch.bfloat16, Kd)
            assert_close(out, ref + bias.double() + res.double(), 2 * rtol, 2 * atol, f"192-row bf16 rep {rep}")


@pytest.mark.parametrize("shape", [(576, 2304, 768), (1088, 3072, 1024), (130, 140, 64), (2176, 1024, 4096)])
def test_gemm_f32_bf16x3(shape):
    """fp32 NT product as ONE bf16 ring-kernel product over [hi|hi|lo] x [hi|lo|hi] (dxa_split3 + epi_f32): 16 mantissa
    bits per operand, so ~1e-5 of the result scale (TF32, what the reference's tf32=True gives, is ~5e-4); the fp32
    epilogue (bias, activation, residual, aux_out, accumulate) is the exact kernel's"""
    M, N, Kd = shape
    a, w = rnd(M, Kd, seed=31), rnd(N, Kd, seed=32, scale=0.1)
    bias, res = rnd(N, seed=33), rnd(M, N, seed=34)
    pre = a.double() @ w.double().t() + bias.double()
    ref = F.gelu(pre, approximate="tanh") + res.double()
    exact = K.mm_nt(a, w, bias=bias, act=2, residual=res)
    aux = torch.empty(M, N, device=DEV)
    with K.f32_gemm_mode("bf16x3"):
        out = K.mm_nt(a, w, bias=bias, act=2, residual=res, aux_out=aux)
        acc = torch.full((M, N), 0.25, device=DEV)
        K.mm_nt(a, w, out=acc, accumulate=True)
        small = K.mm_nt(a[:64], w)                        # below 128 rows: stays on the exact kernels
    assert K.F32_GEMM_MODE == "exact"
    scale = float((a.double() @ w.double().t()).pow(2).mean().sqrt())
    assert float((aux.double() - pre).abs().max()) < 1e-4 * scale
    assert float((out.double() - ref).abs().max()) < 1e-4 * scale
    assert float((acc.double() - 0.25 - (pre - bias.double())).abs().max()) < 1e-4 * scale
    assert float((exact.double() - ref).abs().max()) < 2e-5 * scale
    assert torch.equal(small, K.mm_nt(a[:64], w))
    # the split itself: hi + lo reproduces x to 2^-17 relative
    s3 = K.split3(a, M, Kd, Kd, 0).float()
    assert torch.equal(s3[:, :Kd], s3[:, Kd:2 * Kd])
    assert float(((s3[:, :Kd] + s3[:, 2 * Kd:]) - a).abs().max()) <= float(a.abs().max()) * 2.0 ** -16


@pytest.mark.parametrize("shape", [(1, 768, 64), (17, 100, 768), (36, 3072, 768), (36, 768, 3072), (64, 2304, 768)])
def test_gemm_skinny_f32(shape):
    """fp32 NT with M <= 64 (DiT head at inference): 16-column workgroups, 8-way K split inside the workgroup"""
    M, N, Kd = shape
    a, w = rnd(M, Kd, seed=19), rnd(N, Kd, seed=20, scale=0.1)
    bias, res = rnd(N, seed=21), rnd(M, N, seed=22)
    pre = a.double() @ w.double().t() + bias.double()
    rtol, atol = tol_for(torch.float32, Kd)
    assert_close(K.mm_nt(a, w), a.double() @ w.double().t(), rtol, atol, f"skinny plain {shape}")
    aux = torch.empty(M, N, device=DEV)
    out = K.mm_nt(a, w, bias=bias, act=2, residual=res, aux_out=aux)
    assert_close(aux, pre, rtol, atol, "skinny aux")
    assert_close(out, ACTS[2](pre) + res.double(), 2 * rtol, 2 * atol, f"skinny epilogue {shape}")
    acc = torch.full((M, N), 0.25, device=DEV)
    K.mm_nt(a, w, out=acc, accumulate=True)
    assert_close(acc, a.double() @ w.double().t() + 0.25, rtol, atol, "skinny accumulate")


@pytest.mark.parametrize("shape", [(1, 768, 64), (17, 100, 768), (36, 3072, 768), (63, 2304, 1024), (2, 1000, 3584)])
def test_gemm_skinny_bf16(shape):
    """bf16 NT with M < 64 (KV-cached decode): weight stream, 16-column workgroups, 8-way K split"""
    M, N, Kd = shape
    a = rnd(M, Kd, dtype=torch.bfloat16, seed=23)
    w = rnd(N, Kd, dtype=torch.bfloat16, seed=24, scale=0.1)
    bias, res = rnd(N, dtype=torch.bfloat16, seed=25), rnd(M, N, dtype=torch.bfloat16, seed=26)
    ref = a.double() @ w.double().t()
    rtol, atol = tol_for(torch.bfloat16, Kd)
    assert_close(K.mm_nt(a, w), ref, rtol, atol, f"skinny bf16 plain {shape}")
    out = K.mm_nt(a, w, bias=bias, act=4, residual=res)
    assert_close(out, ACTS[4](ref + bias.double()) + res.double(), 2 * rtol, 2 * atol, f"skinny bf16 epilogue {shape}")
    acc = torch.full((M, N), 0.25, device=DEV, dtype=torch.float32)
    K.mm_nt(a, w, out=acc, accumulate=True)
    assert_close(acc, ref + 0.25, 2e-5, 2e-3 * math.sqrt(Kd / 320), "skinny bf16 f32-out accumulate")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_batched_gqa(dtype):
    B, Hkv, G, S, D = 2, 2, 3, 70, 64
    q = rnd(B, Hkv * G, S, D, dtype=dtype, seed=13)
    k = rnd(B, Hkv, S, D, dtype=dtype, seed=14)
    out = torch.empty(B, Hkv * G, S, S, device=DEV, dtype=torch.float32)
    K.gemm(L.NT, q, k, S, S, D, D, D, out, S, alpha=0.5, nb=(B, Hkv, G),
           sA=(Hkv * G * S * D, G * S * D, S * D), sB=(Hkv * S * D, S * D, 0), sC=(Hkv * G * S * S, G * S * S, S * S))
    ref = 0.5 * torch.einsum("bhgid,bhjd->bhgij", q.double().view(B, Hkv, G, S, D), k.double()).reshape(B, Hkv * G, S, S)
    rtol, atol = (2e-5, 2e-5) if dtype == torch.float32 else (2e-5, 1e-4)
    assert_close(out, ref, rtol, atol, "batched gqa scores")


# ----------------------------------------------------------------------------------------------- norms
@pytest.mark.parametrize("dtype,wdtype", [(torch.float32, torch.float32), (torch.bfloat16, torch.float32),
                                          (torch.bfloat16, torch.bfloat16)])
@pytest.mark.parametrize("rows,cols", [(37, 256), (1030, 3584), (5, 130)])
def test_rmsnorm(dtype, wdtype, rows, cols):
    x = rnd(rows, cols, dtype=dtype, seed=20)
    w = (1 + 0.1 * rnd(cols, seed=21)).to(wdtype)
    dy = rnd(rows, cols, dtype=dtype, seed=22)
    y, rstd = K.rmsnorm_fwd(x, w, 1e-6)
    xr = x.double().requires_grad_(True)
    wr = w.double().requires_grad_(True)
    normed = xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + 1e-6)
    yr = wr * normed
    rtol, atol = (1e-5, 1e-5) if dtype == torch.float32 else (1.0 / 64, 1e-2)
    assert_close(y, yr, rtol, atol, "rmsnorm fwd")
    yr.backward(dy.double())
    dx, dw = K.rmsnorm_bwd(dy, x, w, rstd)
    assert_close(dx, xr.grad, rtol, atol * 2, "rmsnorm dx")
    assert_close(dw, wr.grad, rtol, atol * math.sqrt(rows), "rmsnorm dw")
    # residual branch folded into the kernel: dx + res in one rounding, same weight gradient
    res = rnd(rows, cols, dtype=dtype, seed=28)
    dx2, dw2 = K.rmsnorm_bwd(dy, x, w, rstd, residual=res)
    assert_close(dx2, xr.grad + res.double(), rtol, atol * 2, "rmsnorm dx + residual")
    assert torch.equal(dw2, dw)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("affine", [True, False])
@pytest.mark.parametrize("rows,cols", [(40, 128), (1028, 1024), (7, 70)])
def test_layernorm(dtype, affine, rows, cols):
    x = rnd(rows, cols, dtype=dtype, seed=23) + 0.5
    w = (1 + 0.1 * rnd(cols, seed=24)) if affine else None
    b = 0.1 * rnd(cols, seed=25) if affine else None
    dy = rnd(rows, cols, dtype=dtype, seed=26)
    y, mean, rstd = K.layernorm_fwd(x, w, b, 1e-5)
    xr = x.double().requires_grad_(True)
    wr = w.double().requires_grad_(True) if affine else None
    br = b.double().requires_grad_(True) if affine else None
    yr = F.layer_norm(xr, (cols,), wr, br, 1e-5)
    rtol, atol = (1e-5, 1e-5) if dtype == torch.float32 else (1.0 / 64, 2e-2)
    assert_close(y, yr, rtol, atol, "layernorm fwd")
    yr.backward(dy.double())
    dx, dw, db = K.layernorm_bwd(dy, x, w, mean, rstd)
    assert_close(dx, xr.grad, rtol, atol * 2, "layernorm dx")
    if affine:
        assert_close(dw, wr.grad, rtol, atol * math.sqrt(rows), "layernorm dw")
        assert_close(db, br.grad, rtol, atol * math.sqrt(rows), "layernorm db")
    res = rnd(rows, cols, dtype=dtype, seed=29)
    dx2, dw2, db2 = K.layernorm_bwd(dy, x, w, mean, rstd, residual=res)
    assert_close(dx2, xr.grad + res.double(), rtol, atol * 2, "layernorm dx + residual")
    if affine:
        assert torch.equal(dw2, dw) and torch.equal(db2, db)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,cols", [(4592, 520), (3, 70), (300, 4608)])
def test_colsum(dtype, rows, cols):
    x = rnd(rows, cols, dtype=dtype, seed=27)
    out = K.colsum(x)
    assert_close(out, x.double().sum(0), 1e-5, 1e-4 * math.sqrt(rows), "colsum")
    acc = torch.ones(cols, device=DEV)
    K.colsum(x, out=acc, accumulate=True)
    assert_close(acc, 1 + x.double().sum(0), 1e-5, 1e-4 * math.sqrt(rows), "colsum accumulate")


# ------------------------------------------------------------------------------------------------ RoPE
def _rope_tables(S, D, theta):
    inv = 1.0 / (theta ** (torch.arange(0, D, 2, dtype=torch.float32) / D))
    fr = torch.arange(S, dtype=torch.float32)[:, None] * inv[None]
    return fr.cos().to(DEV).contiguous(), fr.sin().to(DEV).contiguous()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_rope_split_merge(dtype):
    B, S, Hq, Hkv, D = 2, 37, 4, 2, 128
    qkv = rnd(B * S, (Hq + 2 * Hkv) * D, dtype=dtype, seed=30)
    cos_t, sin_t = _rope_tables(S, D, 1e6)
    q, k, v = K.rope_split(qkv, cos_t, sin_t, None, B, S, Hq, Hkv, D)
    x = qkv.double().view(B, S, Hq + 2 * Hkv, D)
    cos = torch.cat([cos_t, cos_t], -1).double()[None, :, None]
    sin = torch.cat([sin_t, sin_t], -1).double()[None, :, None]
    rot = lambda t: torch.cat([-t[..., D // 2:], t[..., :D // 2]], -1)
    qr = (x[:, :, :Hq] * cos + rot(x[:, :, :Hq]) * sin).transpose(1, 2)
    kr = (x[:, :, Hq:Hq + Hkv] * cos + rot(x[:, :, Hq:Hq + Hkv]) * sin).transpose(1, 2)
    vr = x[:, :, Hq + Hkv:].transpose(1, 2)
    rtol, atol = (1e-5, 1e-5) if dtype == torch.float32 else (1.0 / 64, 2e-2)
    assert_close(q, qr, rtol, atol, "rope q")
    assert_close(k, kr, rtol, atol, "rope k")
    assert_close(v, vr, 0, 0, "rope v copy")
    # merge is the transpose (adjoint) of split: <split(x), g> == <x, merge(g)>
    gq, gk, gv = rnd(*q.shape, dtype=dtype, seed=31), rnd(*k.shape, dtype=dtype, seed=32), rnd(*v.shape, dtype=dtype, seed=33)
    dqkv = K.rope_merge(gq, gk, gv, cos_t, sin_t, None, B, S, Hq, Hkv, D)
    lhs = (qr * gq.double()).sum() + (kr * gk.double()).sum() + (vr * gv.double()).sum()
    rhs = (qkv.double() * dqkv.double()).sum()
    assert abs(lhs - rhs) < (1e-6 if dtype == torch.float32 else 3e-2) * (abs(lhs) + 100)


# ------------------------------------------------------------------------------------------- attention
def _attn_ref(q, k, v, causal, scale, kv_start, kv_end):
    B, Hq, Sq, D = q.shape
    Hkv, Sk = k.shape[1], k.shape[2]
    G = Hq // Hkv
    kk = k.double().repeat_interleave(G, 1)
    vv = v.double().repeat_interleave(G, 1)
    s = q.double() @ kk.transpose(-1, -2) * scale
    j = torch.arange(Sk, device=q.device)[None, None, None, :]
    i = torch.arange(Sq, device=q.device)[None, None, :, None]
    vis = torch.ones(B, 1, Sq, Sk, dtype=torch.bool, device=q.device)
    if kv_start is not None:
        vis = vis & (j >= kv_start.view(B, 1, 1, 1))
    if kv_end is not None:
        vis = vis & (j < kv_end.view(B, 1, 1, 1))
    if causal:
        vis = vis & (j <= i + (Sk - Sq))
    s = s.masked_fill(~vis, float("-inf"))
    p = torch.softmax(s, -1)
    p = torch.nan_to_num(p, nan=0.0)
    return p @ vv, torch.logsumexp(s, -1)


ATTN_CASES = [
    # B, Hq, Hkv, Sq, D, causal, padded, head_major
    (2, 4, 2, 150, 128, True, True, True),     # LLM-like GQA causal + right padding
    (2, 2, 2, 257, 64, False, False, False),   # CLIP-like, token-major fused qkv
    (3, 2, 2, 17, 64, False, False, False),    # DiT-like
    (1, 7, 1, 287, 128, True, False, True),    # 7:1 GQA
    (2, 2, 1, 64, 128, True, True, True),
    (2, 8, 1, 200, 256, False, True, True),    # Gemma-like MQA, head_dim 256 (pi0)
    (1, 4, 1, 70, 256, True, False, True),
    (2, 3, 3, 256, 72, False, False, False),   # SigLIP-So400m head width, token-major fused qkv: 72 real columns on 128-wide tiles
    (2, 4, 2, 150, 72, True, True, True),      # ... with GQA, causal and right padding (every mask path at DV < D)
    (8, 16, 16, 256, 72, False, False, False), # enough 128-query tiles to fill the chip: the 8-wave forward / dQ kernels at DV 72
]


@pytest.mark.parametrize("dtype,force_generic", [(torch.bfloat16, False), (torch.bfloat16, True), (torch.float32, True),
                                                 (torch.float32, 2), (torch.bfloat16, 2),       # 2: the materialised forward
                                                 (torch.float32, False)])     # fp32 default dispatch: the head-sized one-launch kernels (DiT-like case)
@pytest.mark.parametrize("case", ATTN_CASES)
def test_attention_fwd_bwd(dtype, force_generic, case):
    B, Hq, Hkv, S, D, causal, padded, head_major = case
    scale = D ** -0.5
    if head_major:
        q, k, v = rnd(B, Hq, S, D, dtype=dtype, seed=40), rnd(B, Hkv, S, D, dtype=dtype, seed=41), rnd(B, Hkv, S, D, dtype=dtype, seed=42)
        o_store = torch.empty(B, S, Hq, D, device=DEV, dtype=dtype)       # token-major output (o_proj input)
        o = o_store.permute(0, 2, 1, 3)
        do_store = rnd(B, Hq, S, D, dtype=dtype, seed=43)
        do = do_store
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    else:
        assert Hq == Hkv
        qkv = rnd(B, S, 3, Hq, D, dtype=dtype, seed=44)
        q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
        o_store = torch.empty(B, S, Hq, D, device=DEV, dtype=dtype)
        o = o_store.permute(0, 2, 1, 3)
        do = rnd(B, S, Hq, D, dtype=dtype, seed=45).permute(0, 2, 1, 3)
        dqkv = torch.empty_like(qkv)
        dq, dk, dv = (dqkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    kv_end = None
    if padded:
        kv_end = torch.tensor([S - 5 * (b + 1) for b in range(B)], dtype=torch.int32, device=DEV)
    lse = K.attn_fwd(q, k, v, o, causal=causal, scale=scale, kv_end=kv_end, force_generic=force_generic)
    ref_o, ref_lse = _attn_ref(q, k, v, causal, scale, None, kv_end)
    rtol, atol = (2e-5, 2e-5) if dtype == torch.float32 else (1.0 / 64, 2e-2)
    assert_close(o, ref_o, rtol, atol, "attn o")
    assert_close(lse, ref_lse, 1e-4 if dtype == torch.float32 else 1e-2, 1e-4 if dtype == torch.float32 else 3e-2, "attn lse")
    # backward vs autograd
    qr, kr, vr = (t.double().detach().clone().requires_grad_(True) for t in (q, k, v))
    ro, _ = _attn_ref(qr, kr, vr, causal, scale, None, kv_end)
    ro.backward(do.double())
    K.attn_bwd(q, k, v, o, lse, do, dq, dk, dv, causal=causal, scale=scale, kv_end=kv_end, force_generic=force_generic)
    rt, at = (1e-4, 1e-4) if dtype == torch.float32 else (1.0 / 32, 6e-2)
    assert_close(dq, qr.grad, rt, at, "attn dq")
    assert_close(dk, kr.grad, rt, at * 2, "attn dk")
    assert_close(dv, vr.grad, rt, at * 2, "attn dv")


@pytest.mark.parametrize("D,Hq,Hkv,S,causal", [(64, 2, 2, 130, False), (128, 4, 2, 150, True), (256, 4, 1, 200, False)])
def test_attention_forward_two_flash_kernels_agree_bit_for_bit(D, Hq, Hkv, S, causal):
    """Round 5: the flash forward runs on row-major V tiles + transposing LDS reads (attn_fwd_tr_k, 16-byte V rows); a V that is
    only 8-byte aligned falls back to the kernel of rounds 1-4 (attn_fwd_flash_k, V transposed through registers).  Both add
    every output element in the same order: the same numbers, bit for bit, and both within the bf16 bound of the fp64 reference"""
    B = 2
    scale = D ** -0.5
    q, k = rnd(B, Hq, S, D, dtype=torch.bfloat16, seed=70), rnd(B, Hkv, S, D, dtype=torch.bfloat16, seed=71)
    v = rnd(B, Hkv, S, D, dtype=torch.bfloat16, seed=72)
    kv_end = torch.tensor([S, S - 9], dtype=torch.int32, device=DEV)
    o1 = torch.empty_like(q)
    lse1 = K.attn_fwd(q, k, v, o1, causal=causal, scale=scale, kv_end=kv_end)
    buf = torch.empty(v.numel() + 4, device=DEV, dtype=torch.bfloat16)
    v8 = buf[4:].view_as(v)                                   # 8-byte aligned, not 16
    v8.copy_(v)
    assert v8.data_ptr() % 16 == 8
    o2 = torch.empty_like(q)
    lse2 = K.attn_fwd(q, k, v8, o2, causal=causal, scale=scale, kv_end=kv_end)
    assert torch.equal(o1, o2) and torch.equal(lse1, lse2)
    ref_o, ref_lse = _attn_ref(q, k, v, causal, scale, None, kv_end)
    assert_close(o1, ref_o, 1.0 / 64, 2e-2, "attn o")
    assert_close(lse1, ref_lse, 1e-2, 3e-2, "attn lse")


def test_attention_left_padding_and_flash_vs_generic():
    B, Hq, Hkv, S, D = 2, 4, 4, 200, 64
    q, k, v = (rnd(B, Hq, S, D, dtype=torch.bfloat16, seed=s) for s in (50, 51, 52))
    o1, o2 = torch.empty_like(q), torch.empty_like(q)
    ks = torch.tensor([0, 37], dtype=torch.int32, device=DEV)
    l1 = K.attn_fwd(q, k, v, o1, causal=True, scale=0.125, kv_start=ks)
    l2 = K.attn_fwd(q, k, v, o2, causal=True, scale=0.125, kv_start=ks, force_generic=True)
    assert_close(o1, o2, 1.0 / 64, 1e-2, "flash vs generic")
    assert_close(l1, l2, 1e-3, 1e-3, "flash vs generic lse")
    assert torch.all(o1[1, :, :37] == 0)         # rows with no visible key
    # fused flash backward vs the GEMM-composed backward under the same left-padding mask
    do = rnd(B, Hq, S, D, dtype=torch.bfloat16, seed=53)
    g1 = [torch.empty_like(q) for _ in range(3)]
    g2 = [torch.empty_like(q) for _ in range(3)]
    K.attn_bwd(q, k, v, o1, l1, do, *g1, causal=True, scale=0.125, kv_start=ks)
    K.attn_bwd(q, k, v, o1, l1, do, *g2, causal=True, scale=0.125, kv_start=ks, force_generic=True)
    for a, b, name in zip(g1, g2, ("dq", "dk", "dv")):
        assert_close(a, b, 1.0 / 32, 6e-2, f"flash bwd vs generic {name}")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_attention_block_prefix_mask_hd256(dtype):
    """pi0's mixture-of-transformers attention: MQA, head_dim 256, per-query key limits (block prefix) and
    per-key validity; forward and backward against an explicit additive-mask reference"""
    B, Hq, Hkv, Sq, Sk, D = 2, 4, 1, 23, 23, 256
    q, k, v = rnd(B, Hq, Sq, D, dtype=dtype, seed=140), rnd(B, Hkv, Sk, D, dtype=dtype, seed=141), rnd(B, Hkv, Sk, D, dtype=dtype, seed=142)
    do = rnd(B, Hq, Sq, D, dtype=dtype, seed=143)
    valid = torch.ones(B, Sk, dtype=torch.bool, device=DEV)
    valid[0, 5:9] = False
    valid[1, 14] = False
    ar = torch.zeros(Sk, dtype=torch.long, device=DEV)
    ar[16] = 1
    ar[17] = 1                                                      # blocks: [0,16) | {16} | [17,23)
    cum = torch.cumsum(ar, 0)
    lim = (cum[None, :] <= cum[:, None]).sum(1).to(torch.int32)     # keys with cumsum <= the query's
    q_limit = lim[None].expand(B, Sq).contiguous()
    mask = (cum[None, :] <= cum[:, None])[None] & valid[:, None, :]                     # [B,Sq,Sk]
    scale = D ** -0.5
    o = torch.empty_like(q)
    lse = K.attn_fwd(q, k, v, o, causal=False, scale=scale, q_limit=q_limit, key_valid=valid.to(torch.uint8))
    qr, kr, vr = (t.double().detach().clone().requires_grad_(True) for t in (q, k, v))
    s = torch.einsum("bhid,bhjd->bhij", qr, kr.repeat_interleave(Hq // Hkv, 1)) * scale
    s = s.masked_fill(~mask[:, None], float("-inf"))
    ref = torch.softmax(s, -1) @ vr.repeat_interleave(Hq // Hkv, 1)
    rtol, atol = (2e-5, 2e-5) if dtype == torch.float32 else (1.0 / 64, 2e-2)
    assert_close(o, ref, rtol, atol, "masked attn o")
    ref.backward(do.double())
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    K.attn_bwd(q, k, v, o, lse, do, dq, dk, dv, causal=False, scale=scale, q_limit=q_limit, key_valid=valid.to(torch.uint8))
    rt, at = (1e-4, 1e-4) if dtype == torch.float32 else (1.0 / 32, 8e-2)
    assert_close(dq, qr.grad, rt, at, "masked attn dq")
    assert_close(dk, kr.grad, rt, at * 2, "masked attn dk")
    assert_close(dv, vr.grad, rt, at * 2, "masked attn dv")
    assert torch.all(dk[0, :, 5:9] == 0) and torch.all(dv[1, :, 14] == 0)


# ----------------------------------------------------------------------------------------- elementwise
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_swiglu_and_acts(dtype):
    rows, Fd = 77, 264
    gu = rnd(rows, 2 * Fd, dtype=dtype, seed=60)
    dout = rnd(rows, Fd, dtype=dtype, seed=61)
    out = K.swiglu_fwd(gu)
    gr = gu.double().requires_grad_(True)
    ref = F.silu(gr[:, :Fd]) * gr[:, Fd:]
    rtol, atol = (1e-5, 1e-5) if dtype == torch.float32 else (1.0 / 64, 1e-2)
    assert_close(out, ref, rtol, atol, "swiglu fwd")
    ref.backward(dout.double())
    assert_close(K.swiglu_bwd(gu, dout), gr.grad, rtol, atol, "swiglu bwd")
    for act in (2, 4):                                       # GeGLU (Gemma) and SiLU through the generic GLU kernels
        g2 = gu.double().detach().requires_grad_(True)
        ref2 = ACTS[act](g2[:, :Fd]) * g2[:, Fd:]
        assert_close(K.glu_fwd(gu, act), ref2, rtol, atol, f"glu fwd {act}")
        ref2.backward(dout.double())
        assert_close(K.glu_bwd(gu, dout, act), g2.grad, rtol, atol, f"glu bwd {act}")
    for act in (1, 2, 3, 4, 5, 6):
        x = rnd(rows, Fd, dtype=dtype, seed=62)
        xr = x.double().requires_grad_(True)
        yr = ACTS[act](xr)
        assert_close(K.act_fwd(x, act), yr, rtol, atol, f"act {act}")
        yr.backward(dout.double())
        assert_close(K.act_bwd(x, dout, act), xr.grad, rtol, atol, f"act bwd {act}")
    a, b = rnd(5, 70, dtype=dtype, seed=63), rnd(5, 70, dtype=dtype, seed=64)
    assert_close(K.add(a, b), a.double() + b.double(), rtol, atol, "add")
    assert_close(K.axpby(a, b, 0.5, -2.0), 0.5 * a.double() - 2.0 * b.double(), rtol, atol, "axpby")
    assert_close(K.mul(a, b), a.double() * b.double(), rtol, atol, "mul")
    x3, g2 = rnd(3, 11, 70, dtype=dtype, seed=66), rnd(3, 70, dtype=dtype, seed=67)
    assert_close(K.mul_rows(x3, g2), x3.double() * g2.double()[:, None, :], rtol, atol, "mul_rows")
    c = K.cast(rnd(33, 12, seed=65), torch.bfloat16)
    assert torch.equal(c, rnd(33, 12, seed=65).to(torch.bfloat16))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_splice_gather(dtype):
    V, d, nimg = 50, 136, 6
    embed = rnd(V, d, dtype=dtype, seed=70)
    img = rnd(2 * nimg, d, dtype=dtype, seed=71)
    PAD = np.iinfo(np.int64).min
    plan = torch.tensor([3, -1, -2, -3, -4, -5, -6, 7, 3, PAD,
                         9, -7, -8, -9, -10, -11, -12, 1, PAD, PAD], dtype=torch.int64, device=DEV)
    out = K.splice_fwd(plan, embed, img)
    ref = torch.zeros(20, d, device=DEV, dtype=dtype)
    for r, p in enumerate(plan.tolist()):
        if p >= 0:
            ref[r] = embed[p]
        elif p != PAD:
            ref[r] = img[-1 - p]
    assert torch.equal(out, ref)
    dout = rnd(20, d, dtype=dtype, seed=72)
    d_embed = torch.zeros(V, d, device=DEV)
    d_img = torch.zeros_like(img)
    K.splice_bwd(plan, dout, d_embed, d_img)
    re = torch.zeros(V, d, device=DEV, dtype=torch.float64)
    ri = torch.zeros(2 * nimg, d, device=DEV, dtype=dtype)
    for r, p in enumerate(plan.tolist()):
        if p >= 0:
            re[p] += dout[r].double()
        elif p != PAD:
            ri[-1 - p] = dout[r]
    assert_close(d_embed, re, 1e-6, 1e-6, "splice d_embed")
    assert torch.equal(d_img, ri)
    # duplicates are summed in a fixed order: bitwise reproducible, and the sparse re-zero clears exactly the touched rows
    for _ in range(3):
        d2 = torch.zeros(V, d, device=DEV)
        K.splice_bwd(plan, dout, d2, None)
        assert torch.equal(d2, d_embed)
    K.zero_rows(plan, d_embed)
    assert d_embed.abs().sum().item() == 0.0
    idx = torch.tensor([4, 19, 4], dtype=torch.int64, device=DEV)
    gth = K.gather_rows(out, idx, torch.float32)
    assert torch.equal(gth, out[idx].float())
    sc = K.scatter_rows(gth, idx, 20, dtype)
    rs = torch.zeros(20, d, device=DEV, dtype=torch.float64)
    rs.index_add_(0, idx, gth.double())
    assert_close(sc, rs, 1.0 / 128, 1e-6, "scatter rows")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_vit_frontend(dtype):
    N, H, P, Cc = 3, 56, 14, 72
    img = rnd(N, 3, H, H, seed=80)
    ld = 592
    rows = K.im2col(img, P, ld, dtype)
    ref = F.unfold(img, P, stride=P).transpose(1, 2).reshape(-1, 3 * P * P)
    assert torch.equal(rows[:, :588], ref.to(dtype))
    assert torch.all(rows[:, 588:] == 0)
    np_ = (H // P) ** 2
    patch = rnd(N * np_, Cc, dtype=dtype, seed=81)
    cls, pos = rnd(Cc, seed=82), rnd(np_ + 1, Cc, seed=83)
    x = K.vit_embed_fwd(patch, cls, pos, N, np_)
    ref = torch.cat([cls.to(dtype).expand(N, 1, Cc), patch.view(N, np_, Cc)], 1).float() + pos
    assert_close(x, ref, 1e-6 if dtype == torch.float32 else 1.0 / 128, 1e-6, "vit embed")
    dx = rnd(N, np_ + 1, Cc, dtype=dtype, seed=84)
    assert torch.equal(K.vit_embed_bwd(dx, N, np_), dx[:, 1:].reshape(-1, Cc))


def test_diffusion_glue():
    N, T, A, hd, d = 8, 16, 7, 96, 136
    x0, noise = rnd(N, T, A, seed=90), rnd(N, T, A, seed=91)
    a, s = torch.rand(N, device=DEV), torch.rand(N, device=DEV)
    assert_close(K.qsample(x0, noise, a, s), a[:, None, None] * x0 + s[:, None, None] * noise, 1e-6, 1e-7, "qsample")
    t = torch.tensor([0, 1, 5, 17, 50, 63, 98, 99], dtype=torch.float32, device=DEV)
    half = 128
    freqs = torch.exp(-math.log(10000) * torch.arange(0, half, dtype=torch.float32) / half).to(DEV)
    emb = K.timestep_embedding(t, freqs)
    args = t[:, None] * freqs[None]
    assert_close(emb, torch.cat([args.cos(), args.sin()], -1), 1e-5, 2e-6, "timestep embedding")
    xe, te, ze, pos = rnd(N, T, hd, seed=92), rnd(N, hd, seed=93), rnd(N, hd, seed=94), rnd(T + 1, hd, seed=95)
    h = K.dit_assemble_fwd(xe, te, ze, pos)
    ref = torch.cat([(te + ze)[:, None], xe], 1) + pos
    assert_close(h, ref, 1e-6, 1e-6, "dit assemble")
    dh = rnd(N, T + 1, hd, seed=96)
    dxe, dc = K.dit_assemble_bwd(dh)
    assert torch.equal(dxe, dh[:, 1:]) and torch.equal(dc, dh[:, 0])
    z, unc = rnd(N, d, seed=97), rnd(d, seed=98)
    drop = torch.tensor([0, 1, 0, 0, 1, 0, 0, 0], dtype=torch.uint8, device=DEV)
    zo = K.token_drop(z, unc, drop)
    assert torch.equal(zo, torch.where(drop.bool()[:, None], unc[None], z))
    dz, dunc = K.token_drop_bwd(z, drop)
    assert torch.equal(dz, torch.where(drop.bool()[:, None], torch.zeros_like(z), z))
    assert_close(dunc, z[drop.bool()].sum(0), 1e-6, 1e-6, "dunc")
    pred, tgt = rnd(N, T, A, seed=99), rnd(N, T, A, seed=100)
    loss, dpred = K.mse_loss(pred, tgt, gscale=0.5)
    assert_close(loss, ((pred - tgt) ** 2).mean().view(1), 1e-5, 1e-6, "mse")
    assert_close(dpred, 0.5 * 2 * (pred - tgt) / pred.numel(), 1e-5, 1e-9, "mse grad")
    # ddim step with CFG
    Bq, per = 2, T * A
    x = rnd(2 * Bq, T, A, seed=101)
    x[Bq:] = x[:Bq]
    mo = rnd(2 * Bq, T, A, seed=102)
    xr = x.clone()
    c1, c2, abp = 1.31, 0.85, 0.77
    K.ddim_step(x, mo, Bq, True, 1.5, c1, c2, abp)
    eps = mo[Bq:] + 1.5 * (mo[:Bq] - mo[Bq:])
    x0p = c1 * xr[:Bq] - c2 * eps
    e2 = (c1 * xr[:Bq] - x0p) / c2
    ref = x0p * math.sqrt(abp) + math.sqrt(1 - abp) * e2
    assert_close(x[:Bq], ref, 1e-5, 1e-6, "ddim step")
    assert torch.equal(x[:Bq], x[Bq:])
    del per


# --------------------------------------------------------------------------------------------- optimizer
def test_adamw_matches_torch():
    torch.manual_seed(0)
    sizes = [1000, 37, 4096 * 3 + 5, 8, 70000]
    n = sum(sizes)
    p = rnd(n, seed=110).contiguous()
    g = rnd(n, seed=111, scale=0.1).contiguous()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    shadow = torch.empty(n, device=DEV, dtype=torch.bfloat16)
    params = [p[sum(sizes[:i]):sum(sizes[:i + 1])].clone().requires_grad_(True) for i in range(len(sizes))]
    groups = [dict(params=[params[0], params[2], params[4]], weight_decay=0.01, lr=1e-3),
              dict(params=[params[1], params[3]], weight_decay=0.0, lr=5e-4)]
    opt = torch.optim.AdamW(groups, betas=(0.9, 0.999), eps=1e-8)
    # chunk table: chunks of <= 4096 elements aligned to tensor boundaries
    cs, cl, cg = [], [], []
    off = 0
    for i, sz in enumerate(sizes):
        grp = 0 if i in (0, 2, 4) else 1
        o = 0
        while o < sz:
            ln = min(4096, sz - o)
            cs.append(off + o); cl.append(ln); cg.append(grp)
            o += ln
        off += sz
    cs_t = torch.tensor(cs, dtype=torch.int64, device=DEV)
    cl_t = torch.tensor(cl, dtype=torch.int32, device=DEV)
    cg_t = torch.tensor(cg, dtype=torch.int32, device=DEV)
    ss = torch.zeros(1, device=DEV)
    scratch = torch.empty(4096, device=DEV, dtype=torch.float64)
    norm, coef = torch.zeros(1, device=DEV), torch.zeros(1, device=DEV)
    for step in (1, 2, 3):
        gs = g * (1 + 0.3 * step)
        for i, q_ in enumerate(params):
            q_.grad = gs[sum(sizes[:i]):sum(sizes[:i + 1])].clone()
        tn = torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step()
        K.sumsq(gs, ss, scratch)
        K.clip_coef(ss, 1.0, norm, coef)
        assert abs(norm.item() - tn.item()) < 1e-4 * tn.item()
        K.adamw(p, gs, m, v, shadow, cs_t, cl_t, cg_t, [1e-3, 5e-4], [0.01, 0.0], 0.9, 0.999, 1e-8, step, clip=coef)
        ref = torch.cat([q_.detach() for q_ in params])
        assert_close(p, ref, 1e-5, 2e-6, f"adamw step {step}")
        assert torch.equal(shadow, p.to(torch.bfloat16))
    x = rnd(1000, seed=112)
    K.scale_(x, 0.25)
    assert_close(x, rnd(1000, seed=112) * 0.25, 0, 0, "scale")
    # bf16 gradient arena (the data-parallel communication copy): same update as with those values widened to fp32
    g16 = gs.to(torch.bfloat16)
    p1, m1, v1 = p.clone(), m.clone(), v.clone()
    p2, m2, v2 = p.clone(), m.clone(), v.clone()
    s1, s2 = torch.empty_like(shadow), torch.empty_like(shadow)
    K.adamw(p1, g16.float(), m1, v1, s1, cs_t, cl_t, cg_t, [1e-3, 5e-4], [0.01, 0.0], 0.9, 0.999, 1e-8, 4, clip=coef)
    K.adamw(p2, g16, m2, v2, s2, cs_t, cl_t, cg_t, [1e-3, 5e-4], [0.01, 0.0], 0.9, 0.999, 1e-8, 4, clip=coef)
    assert torch.equal(p1, p2) and torch.equal(m1, m2) and torch.equal(v1, v2) and torch.equal(s1, s2)
    K.sumsq(g16, ss, scratch)
    assert abs(ss.item() - g16.double().pow(2).sum().item()) < 1e-5 * ss.item()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("R,C", [(4592, 3584), (70, 130), (64, 64), (1, 7), (287, 18944)])
def test_transpose_and_permute(dtype, R, C):
    x = rnd(R, C, dtype=dtype, seed=120)
    y = K.transpose(x, 64)
    Rp = (R + 63) // 64 * 64
    assert y.shape == (C, Rp)
    assert torch.equal(y[:, :R], x.t())
    assert torch.all(y[:, R:] == 0)
    v = rnd(R, C + 8, dtype=dtype, seed=121)[:, 8:]          # strided source view
    assert torch.equal(K.transpose(v, 1), v.t())
    B, S, H, D = 2, 37, 3, 64
    t = rnd(B, S, H, D, dtype=dtype, seed=122)
    hm = K.permute_bshd(t, B, S, H, D, True)
    assert torch.equal(hm, t.permute(0, 2, 1, 3))
    assert torch.equal(K.permute_bshd(hm, B, S, H, D, False), t)


# ------------------------------------------------------------------------------------- LM head (row A10)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,V", [(37, 1000), (5, 152064), (9, 131)])
def test_cross_entropy_and_argmax(dtype, rows, V):
    x = rnd(rows, V, dtype=dtype, seed=130, scale=3.0)
    labels = torch.randint(0, V, (rows,), device=DEV, generator=torch.Generator(device=DEV).manual_seed(5))
    labels[1] = -100
    labels[rows - 1] = -100
    row_loss, lse = K.cross_entropy_fwd(x, labels)
    xr = x.double().requires_grad_(True)
    ref_rows = F.cross_entropy(xr, labels, ignore_index=-100, reduction="none")
    tol = 1e-5 if dtype == torch.float32 else 1e-5       # the kernel computes in fp32 from the stored values
    assert_close(lse, torch.logsumexp(xr, dim=1), tol, tol, "ce lse")
    assert_close(row_loss, ref_rows, tol, 1e-5, "ce row loss")
    n_valid = int((labels != -100).sum())
    g = torch.tensor([0.5], device=DEV)
    (ref_rows.sum() / n_valid * 0.5).backward()
    d = K.cross_entropy_bwd(x, labels, lse, g, 1.0 / n_valid)
    rt, at = (1e-5, 1e-7) if dtype == torch.float32 else (1.0 / 128, 1e-6)
    assert_close(d, xr.grad, rt, at, "ce dlogits")
    assert torch.all(d[1] == 0) and torch.all(d[rows - 1] == 0)
    inplace = x.clone()
    K.cross_entropy_bwd(inplace, labels, lse, g, 1.0 / n_valid, out=inplace)
    assert torch.equal(inplace, d)
    # argmax: first index among ties, any column count
    y = x.clone()
    y[0, 7] = y[0].max() + 1
    y[0, V - 3] = y[0, 7]                                  # tie: index 7 wins
    y[2] = 0                                               # all equal: index 0
    assert torch.equal(K.argmax_rows(y), torch.argmax(y.float(), dim=1))
    assert int(K.argmax_rows(y)[0]) == 7 and int(K.argmax_rows(y)[2]) == 0
    if rows >= 5:                                          # (5, 152064) in bf16: the 1024-thread kernel of the decode step
        y[3, V - 1] = y[3].max() + 2                       # the maximum in the row's last element (the unpaired tail load)
        y[4] = float("-inf")                               # nothing to find: index 0 like torch
        got = K.argmax_rows(y)
        assert int(got[3]) == V - 1 and int(got[4]) == 0 and torch.equal(got, torch.argmax(y.float(), dim=1))


# ------------------------------------------------------------------------------------------------ fused DiT blocks
@pytest.mark.parametrize("offset", [0.0, 50.0])
@pytest.mark.parametrize("cfg", [(2, 17, 768, 12, 3072, 3), (2, 17, 128, 2, 512, 2), (1, 17, 192, 3, 768, 4), (2, 23, 256, 4, 1024, 1)])
def test_dit_blocks_fused(cfg, offset):
    """the persistent DiT-block kernel (one launch, device-wide barriers) against the block arithmetic in fp64.
    offset 50: rows whose mean is 50x their spread — the fused LayerNorm takes its statistics as E[x^2] - mu^2 and applies
    rs * (acc - mu * wsum) in the epilogue; both cancel ~2.5e3 : 1 here, which fp32 accumulation still resolves to ~1e-4 of
    the result scale (the zero-mean case alone could not show that)"""
    N, T1, H, heads, I, depth = cfg
    M = N * T1
    h0 = rnd(M, H, seed=41) + offset
    ws, ptrs = [], []
    for k in range(depth):
        blk = [rnd(3 * H, H, seed=50 + 10 * k, scale=H ** -0.5), rnd(3 * H, seed=51 + 10 * k, scale=0.1),
               rnd(H, H, seed=52 + 10 * k, scale=H ** -0.5), rnd(H, seed=53 + 10 * k, scale=0.1),
               rnd(I, H, seed=54 + 10 * k, scale=H ** -0.5), rnd(I, seed=55 + 10 * k, scale=0.1),
               rnd(H, I, seed=56 + 10 * k, scale=I ** -0.5), rnd(H, seed=57 + 10 * k, scale=0.1)]
        ws.append(blk)
        ptrs += [w.data_ptr() for w in blk]
    assert K.dit_blocks_supported(N, T1, H, heads, I)
    table = torch.tensor(ptrs, dtype=torch.int64).to(DEV)
    ref = h0.double()
    for qw, qb, pw, pb, w1, b1, w2, b2 in ws:
        y = F.layer_norm(ref, (H,), eps=1e-6)
        qkv = (y @ qw.double().t() + qb.double()).view(N, T1, 3, heads, 64)
        q, k_, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
        att = torch.softmax(q @ k_.transpose(-1, -2) / 8.0, dim=-1) @ v
        ref = ref + att.permute(0, 2, 1, 3).reshape(M, H) @ pw.double().t() + pb.double()
        y = F.layer_norm(ref, (H,), eps=1e-6)
        ref = ref + F.gelu(y @ w1.double().t() + b1.double(), approximate="tanh") @ w2.double().t() + b2.double()
    for rep in range(3):                                           # the barrier counter is reset by every call
        out = K.dit_blocks_fwd(h0.clone(), table, depth, N, T1, H, heads, I, 1e-6)
        tol = 2e-4 if offset == 0.0 else 6e-4
        assert_close(out, ref, tol, tol * float(ref.abs().max()), f"fused DiT blocks {cfg} offset {offset} rep {rep}")
    with pytest.raises(L.DxaError):
        K.dit_blocks_fwd(torch.zeros(48, H, device=DEV), table, depth, 2, 24, H, heads, I, 1e-6)     # 48 rows: no spare row for the ones trick


# ------------------------------------------------------------------------------------------------ bf16-operand DiT sampler
def _dit_weights(H, I, depth, seed0=50, per=False):
    """``per``: + per_attn.in_proj_weight / in_proj_bias / out_proj.weight / out_proj.bias / norm3.weight / norm3.bias (MemVLA's block)"""
    ws, ptrs = [], []
    for k in range(depth):
        blk = [rnd(3 * H, H, seed=seed0 + 10 * k, scale=H ** -0.5), rnd(3 * H, seed=seed0 + 1 + 10 * k, scale=0.1),
               rnd(H, H, seed=seed0 + 2 + 10 * k, scale=H ** -0.5), rnd(H, seed=seed0 + 3 + 10 * k, scale=0.1),
               rnd(I, H, seed=seed0 + 4 + 10 * k, scale=H ** -0.5), rnd(I, seed=seed0 + 5 + 10 * k, scale=0.1),
               rnd(H, I, seed=seed0 + 6 + 10 * k, scale=I ** -0.5), rnd(H, seed=seed0 + 7 + 10 * k, scale=0.1)]
        if per:
            blk += [rnd(3 * H, H, seed=seed0 + 1000 + 10 * k, scale=H ** -0.5), rnd(3 * H, seed=seed0 + 1001 + 10 * k, scale=0.1),
                    rnd(H, H, seed=seed0 + 1002 + 10 * k, scale=H ** -0.5), rnd(H, seed=seed0 + 1003 + 10 * k, scale=0.1),
                    (1.0 + rnd(H, seed=seed0 + 1004 + 10 * k, scale=0.2)).contiguous(), rnd(H, seed=seed0 + 1005 + 10 * k, scale=0.2)]
        ws.append(blk)
        ptrs += [w.data_ptr() for w in blk]
    return ws, torch.tensor(ptrs, dtype=torch.int64).to(DEV)


@pytest.mark.parametrize("H,I,depth", [(768, 3072, 2), (192, 768, 3), (128, 512, 1)])
def test_dit_bf16_pack_is_the_rounded_matrix_in_operand_order(H, I, depth):
    """dxa_dit_bf16_pack: every matrix rounded to bf16 (round to nearest even, what torch's .to(bfloat16) does) laid out as
    [16 output columns][K / 32][16][32] tiles, bit for bit; sum_k of the ROUNDED weights for the two LayerNorm-fed matrices;
    the pointer table; re-packing in place after the weights moved"""
    ws, table = _dit_weights(H, I, depth)
    arena, ptab = K.dit_bf16_pack(table, depth, H, I)
    per = arena.numel() // depth
    tab = ptab.cpu().tolist()
    base = arena.data_ptr()

    def check():
        for k, (qw, qb, pw, pb, w1, b1, w2, b2) in enumerate(ws):
            e = tab[10 * k: 10 * k + 10]
            assert [e[1], e[3], e[5], e[7]] == [qb.data_ptr(), pb.data_ptr(), b1.data_ptr(), b2.data_ptr()]
            for w, addr in ((qw, e[0]), (pw, e[2]), (w1, e[4]), (w2, e[6])):
                Nn, Kk = w.shape
                off = addr - base
                assert k * per <= off < (k + 1) * per and off % 256 == 0
                got = arena[off: off + Nn * Kk * 2].view(torch.bfloat16).view(Nn // 16, Kk // 32, 16, 32)
                want = w.to(torch.bfloat16).view(Nn // 16, 16, Kk // 32, 32).permute(0, 2, 1, 3)
                assert torch.equal(got.view(torch.int16), want.contiguous().view(torch.int16))
            for w, addr in ((qw, e[8]), (w1, e[9])):
                off = addr - base
                got = arena[off: off + w.shape[0] * 4].view(torch.float32)
                want = w.to(torch.bfloat16).double().sum(1)
                assert_close(got, want, 1e-5, 1e-5 * float(want.abs().max()), "sum of the rounded weights")
    check()
    for blk in ws:
        for w in blk:
            w.mul_(1.5).add_(0.01)
    K.dit_bf16_pack(table, depth, H, I, out=(arena, ptab))
    assert ptab.cpu().tolist() == tab
    check()


def _dit_sampler_restated(x, ze, te, pos, xw, xb, fw, fb, coef, ws, N, heads, cfg_scale, round_ops, kv=None):
    """float64 restatement of the one-launch sampler (DiT.forward_with_cfg, dit.py:273-311, inside ddim_sample_loop,
    diffusion.py:714-794); round_ops: the products read bf16 weights and bf16 copies of their input activations, LayerNorm is
    folded as rs (bf16(h) W^T - mu sum_k W) like csrc/dit_fused.hip does"""
    R = (lambda t: t.to(torch.bfloat16).double()) if round_ops else (lambda t: t.double())
    x = x.double().clone()
    nb, T, A = x.shape
    H = ze.shape[1]
    ln = lambda h: (h.mean(-1, keepdim=True), (h.var(-1, unbiased=False, keepdim=True) + 1e-6).rsqrt())
    for s in range(te.shape[0]):
        xe = (x @ xw.double().T + xb.double()).repeat(N // nb, 1, 1)
        h = torch.cat([(te[s].double()[None, :] + ze.double())[:, None, :], xe], 1) + pos.double()[None]
        for bi, (qw, qb, pw, pb, w1, b1, w2, b2, *px) in enumerate(ws):
            mu, rs = ln(h)
            Wq = R(qw)
            qkv = rs * (R(h) @ Wq.T - mu * Wq.sum(1)) + qb.double()
            q, kk, v = qkv.reshape(N, T + 1, 3, heads, 64).permute(2, 0, 3, 1, 4)
            o = (torch.softmax((q @ kk.transpose(-1, -2)) * 0.125, -1) @ v).permute(0, 2, 1, 3).reshape(N, T + 1, H)
            h = h + R(o) @ R(pw).T + pb.double()
            if px:
                # x + MHA(norm3 x, per, per) (memvla/action_model/dit.py:158-185): the query rows of the packed in_proj with norm3's
                # affine folded in BEFORE the rounding, keys / values = the request's cached projections kv [depth, N, P, 2, H]
                iw, ib, ow, ob_, g3, b3 = px
                mu, rs = ln(h)
                Wf = R(iw[:H].double() * g3.double()[None, :])
                q2 = rs * (R(h) @ Wf.T - mu * Wf.sum(1)) + (ib[:H].double() + iw[:H].double() @ b3.double())
                P_ = kv.shape[2]
                q2 = q2.reshape(N, T + 1, heads, 64).permute(0, 2, 1, 3)
                k2 = kv[bi, :, :, 0].double().reshape(N, P_, heads, 64).permute(0, 2, 1, 3)
                v2 = kv[bi, :, :, 1].double().reshape(N, P_, heads, 64).permute(0, 2, 1, 3)
                o2 = (torch.softmax((q2 @ k2.transpose(-1, -2)) * 0.125, -1) @ v2).permute(0, 2, 1, 3).reshape(N, T + 1, H)
                h = h + R(o2) @ R(ow).T + ob_.double()
            mu, rs = ln(h)
            W1 = R(w1)
            a = F.gelu(rs * (R(h) @ W1.T - mu * W1.sum(1)) + b1.double(), approximate="tanh")
            h = h + R(a) @ R(w2).T + b2.double()
        mu, rs = ln(h)
        eps = (((h - mu) * rs) @ fw.double().T + fb.double())[:, 1:, :]
        if cfg_scale is not None:
            eps = eps[nb:] + cfg_scale * (eps[:nb] - eps[nb:])
        c0, c1, ab = (coef[s, i].double() for i in range(3))
        x0 = c0 * x - c1 * eps
        x = x0 * ab.sqrt() + (1 - ab).sqrt() * ((c0 * x - x0) / c1)
    return x


@pytest.mark.parametrize("cfg", [(768, 12, 3072, 1, 1, True), (768, 12, 3072, 12, 10, True), (192, 3, 768, 3, 10, False), (128, 2, 512, 2, 4, True)])
def test_dit_sample_bf16_operands(cfg):
    """dxa_dit_sample_bf16_fwd (the sampler of a model served in bfloat16: bf16 MFMA operands, fp32 residual stream / LayerNorm /
    attention / accumulation) against the fp64 restatement of ITS arithmetic.  One block and one step: 5e-4 (fp32 accumulation and a rounding flip or two; measured 1.5e-4
    accumulation separates them).  At depth: two evaluations of the same bf16-operand arithmetic that differ by 1e-7 somewhere round
    a few hundred of the 10^7 operand elements to DIFFERENT bf16 neighbours, and each flip is worth 4e-3 of that element — they end
    ~1e-3 apart, as far as either is from the exact result; bound 6e-3, and within 8e-3 of the exact fp64 sampler (the reference's
    own bf16 head sits 6e-3 from its fp32 head over the whole request, tests/golden/cogact_depth28_ref.npz).  Run-to-run bit-identical."""
    H, heads, I, depth, steps, use_cfg = cfg
    T, A, nb = 16, 7, 1
    N = 2 if use_cfg else 1
    ws, table = _dit_weights(H, I, depth, seed0=150)
    x0 = rnd(nb, T, A, seed=7)
    ze, te, pos = rnd(N, H, seed=8, scale=0.5), rnd(steps, H, seed=9, scale=0.5), rnd(T + 1, H, seed=10, scale=0.1)
    xw, xb, fw, fb = rnd(H, A, seed=11, scale=0.3), rnd(H, seed=12, scale=0.1), rnd(A, H, seed=13, scale=H ** -0.5), rnd(A, seed=14, scale=0.1)
    ab = torch.linspace(0.05, 0.95, steps + 1, device=DEV, dtype=torch.float64)
    coef = torch.zeros(steps, 4, device=DEV)
    coef[:, 0] = (1.0 / ab[:-1]).sqrt().float(); coef[:, 1] = (1.0 / ab[:-1] - 1.0).sqrt().float(); coef[:, 2] = ab[1:].float()
    assert K.dit_sample_bf16_supported(N, T + 1, H, heads, I)
    arena, ptab = K.dit_bf16_pack(table, depth, H, I)
    outs = [K.dit_sample_bf16_fwd(x0.clone(), ze, te, pos, xw, xb, fw, fb, coef, nb, use_cfg, 1.5, ptab, depth, T + 1, H, heads, I, 1e-6)
            for _ in range(3)]
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert not K.dit_blocks_timed_out()
    want = _dit_sampler_restated(x0, ze, te, pos, xw, xb, fw, fb, coef, ws, N, heads, 1.5 if use_cfg else None, True)
    exact = _dit_sampler_restated(x0, ze, te, pos, xw, xb, fw, fb, coef, ws, N, heads, 1.5 if use_cfg else None, False)
    scale = float(exact.abs().max())
    d_same, d_exact = float((outs[0].double() - want).abs().max()) / scale, float((outs[0].double() - exact).abs().max()) / scale
    print(f"bf16-operand sampler {cfg}: vs its restatement {d_same:.2e}, vs exact {d_exact:.2e}")
    assert d_same < (5e-4 if depth * steps == 1 else 6e-3), d_same
    assert d_exact < 8e-3, d_exact


@pytest.mark.parametrize("cfg", [(1024, 16, 4096, 1, 1, True, 256), (1024, 16, 4096, 24, 10, True, 256), (192, 3, 768, 3, 4, False, 64),
                                 (128, 2, 512, 2, 3, True, 128)])
def test_dit_sample_bf16_with_perceptual_attention(cfg):
    """dxa_dit_sample_bf16_per_fwd — MemVLA's DiT (x + attn(norm1 x); x + MHA(norm3 x, per, per); x + mlp(norm2 x),
    memvla/action_model/dit.py:136-185) sampled in one launch — against the fp64 restatement of its arithmetic and of the exact
    sampler, at DiT-L size (1024 wide, 24 blocks, 256 perceptual keys, 10 steps) and at small shapes; the packed query matrix is the
    rounded W diag(gamma3) with b + W beta3 beside it; run-to-run bit-identical"""
    H, heads, I, depth, steps, use_cfg, P_ = cfg
    T, A, nb = 16, 7, 1
    N = 2 if use_cfg else 1
    ws, table = _dit_weights(H, I, depth, seed0=350, per=True)
    x0 = rnd(nb, T, A, seed=7)
    ze, te, pos = rnd(N, H, seed=8, scale=0.5), rnd(steps, H, seed=9, scale=0.5), rnd(T + 1, H, seed=10, scale=0.1)
    xw, xb, fw, fb = rnd(H, A, seed=11, scale=0.3), rnd(H, seed=12, scale=0.1), rnd(A, H, seed=13, scale=H ** -0.5), rnd(A, seed=14, scale=0.1)
    kv = rnd(depth, N, P_, 2, H, seed=15, scale=1.0)
    ab = torch.linspace(0.05, 0.95, steps + 1, device=DEV, dtype=torch.float64)
    coef = torch.zeros(steps, 4, device=DEV)
    coef[:, 0] = (1.0 / ab[:-1]).sqrt().float(); coef[:, 1] = (1.0 / ab[:-1] - 1.0).sqrt().float(); coef[:, 2] = ab[1:].float()
    assert K.dit_sample_bf16_supported(N, T + 1, H, heads, I, P_)
    arena, ptab = K.dit_bf16_pack(table, depth, H, I, per=True)
    # the folded query matrix and bias of block 0
    e = ptab.cpu().tolist()[:16]
    iw, ib, g3, b3 = ws[0][8], ws[0][9], ws[0][12], ws[0][13]
    off = e[10] - arena.data_ptr()
    got = arena[off: off + H * H * 2].view(torch.bfloat16).view(H // 16, H // 32, 16, 32)
    want = (iw[:H] * g3[None, :]).to(torch.bfloat16).view(H // 16, 16, H // 32, 32).permute(0, 2, 1, 3)
    assert torch.equal(got.view(torch.int16), want.contiguous().view(torch.int16))
    off = e[11] - arena.data_ptr()
    assert_close(arena[off: off + H * 4].view(torch.float32), ib[:H].double() + iw[:H].double() @ b3.double(), 1e-5, 1e-5, "folded bias")
    assert e[14] == ws[0][11].data_ptr()
    outs = [K.dit_sample_bf16_fwd(x0.clone(), ze, te, pos, xw, xb, fw, fb, coef, nb, use_cfg, 1.5, ptab, depth, T + 1, H, heads, I, 1e-6,
                                  per_kv=kv) for _ in range(3)]
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert not K.dit_blocks_timed_out()
    want = _dit_sampler_restated(x0, ze, te, pos, xw, xb, fw, fb, coef, ws, N, heads, 1.5 if use_cfg else None, True, kv=kv)
    exact = _dit_sampler_restated(x0, ze, te, pos, xw, xb, fw, fb, coef, ws, N, heads, 1.5 if use_cfg else None, False, kv=kv)
    scale = float(exact.abs().max())
    d_same, d_exact = float((outs[0].double() - want).abs().max()) / scale, float((outs[0].double() - exact).abs().max()) / scale
    print(f"bf16-operand sampler with perceptual attention {cfg}: vs its restatement {d_same:.2e}, vs exact {d_exact:.2e}")
    assert d_same < (5e-4 if depth * steps == 1 else 8e-3), d_same
    assert d_exact < 1.2e-2, d_exact


# --------------------------------------------------------------- sum(g^2) out of the dW product's epilogue
@pytest.mark.parametrize("M,N,Kd,accum", [(512, 768, 300, False), (770, 520, 4592, True), (3584, 4608, 1024, False),
                                         (70, 50, 96, False)])
def test_gemm_sumsq_epilogue(M, N, Kd, accum):
    """dxa_gemm_desc.sumsq: per-tile partials whose plain sum is sum(C^2) of the FINAL C (after accumulate); every slot is
    written; reproducible bit for bit; kernels without that epilogue (last shape: generic tile kernel) take the fallback"""
    dy = rnd(Kd, M, dtype=torch.bfloat16, seed=71)
    x = rnd(Kd, N, dtype=torch.bfloat16, seed=72)
    slots = K.gemm_sumsq_slots(M, N)
    assert slots == ((M + 255) // 256) * ((N + 255) // 256)
    outs = []
    for rep in range(2):
        out = rnd(M, N, seed=73).contiguous() if accum else torch.empty(M, N, device=DEV)
        part = torch.full((slots,), float("nan"), device=DEV)
        K.mm_tn(dy, x, out=out, accumulate=accum, sumsq=part)
        assert torch.isfinite(part).all()
        ref = out.double().pow(2).sum().item()
        assert abs(part.double().sum().item() - ref) <= 1e-5 * ref
        tot = torch.ones(1, device=DEV)
        K.sum_f32(part, tot, accumulate=True)
        assert abs(tot.item() - 1.0 - ref) <= 1e-5 * ref
        outs.append((out, part))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    with pytest.raises(Exception):
        K.mm_tn(dy, x, out=torch.empty(M, N, device=DEV), sumsq=torch.empty(slots + 1, device=DEV))


def test_sumsq_ranges():
    base = rnd(200000, seed=74)
    starts = torch.tensor([0, 64, 1000, 150000, 199999], dtype=torch.int64, device=DEV)
    lens = torch.tensor([10, 700, 0, 40001, 1], dtype=torch.int64, device=DEV)
    scratch = torch.empty(4096, device=DEV, dtype=torch.float64)
    out = torch.full((1,), 2.0, device=DEV)
    K.sumsq_ranges(base, starts, lens, out, scratch, accumulate=True)
    ref = sum(base[a:a + n].double().pow(2).sum().item() for a, n in zip(starts.tolist(), lens.tolist()))
    assert abs(out.item() - 2.0 - ref) <= 1e-6 * ref
    hb = base.bfloat16()
    K.sumsq_ranges(hb, starts, lens, out, scratch)
    ref = sum(hb[a:a + n].double().pow(2).sum().item() for a, n in zip(starts.tolist(), lens.tolist()))
    assert abs(out.item() - ref) <= 1e-6 * ref


# ------------------------------------------------------------------- few-row NT kernel (128x128 tiles, M <= 1024)
@pytest.mark.parametrize("shape", [(543, 4608, 3584), (543, 3584, 18944), (514, 1024, 4096), (514, 4096, 1024), (64, 64, 64),
                                   (130, 300, 64), (1024, 200, 128), (700, 136, 192), (543, 37888, 3584), (514, 1024, 1024),
                                   (640, 1000, 2048), (400, 1536, 3072)])
def test_gemm_few_rows_t128(shape):
    """bf16 NT products on a few hundred rows (batch-1 prefill S = 543, ViT on 2 x 257 tokens): 128x128-tile LDS-DMA ring
    kernel; 1, 2, 3, 4 and many K tiles (the ring has 4 stages), split-K over 2 / 3 / 4 slices where few tiles meet a deep K
    (40 tiles x K 4096 / 1024, 40 x 2048, 48 x 3072), ragged M / N, the whole epilogue menu (bias, residual,
    activation + pre-activation copy, activation gradient, fp32 accumulate), every element against fp64, repetitions bitwise"""
    M, N, Kd = shape
    a, w = rnd(M, Kd, dtype=torch.bfloat16, seed=91), rnd(N, Kd, dtype=torch.bfloat16, seed=92, scale=0.1)
    ref = a.double() @ w.double().t()
    rtol, atol = tol_for(torch.bfloat16, Kd)
    bias, res = rnd(N, dtype=torch.bfloat16, seed=93), rnd(M, N, dtype=torch.bfloat16, seed=94)
    first = None
    for rep in range(3):
        out = K.mm_nt(a, w, bias=bias, residual=res)
        assert_close(out, ref + bias.double() + res.double(), 2 * rtol, 2 * atol, f"t128 bias+residual rep {rep}")
        if first is None:
            first = out.clone()
        assert torch.equal(out, first), f"t128 result changed between repetitions (rep {rep})"
    out32 = torch.full((M, N), 0.5, device=DEV, dtype=torch.float32)
    K.mm_nt(a, w, out=out32, accumulate=True)
    assert_close(out32, ref + 0.5, 2e-5, 2e-3 * math.sqrt(max(Kd, 320) / 320), "t128 f32 accumulate")
    if N <= 8192:
        pre = torch.empty((M, N), device=DEV, dtype=torch.bfloat16)
        y = K.mm_nt(a, w, bias=bias, act=3, aux_out=pre)                       # CLIP fc1: quick_gelu, pre-activation saved
        assert_close(pre, ref + bias.double(), 2 * rtol, 2 * atol, "t128 aux")
        assert_close(y, ACTS[3](ref + bias.double()), 2 * rtol, 2 * atol, "t128 act")
        g = K.mm_nt(a, w, mulgrad=res, act=3)
        xg = res.double().requires_grad_(True)
        ACTS[3](xg).backward(ref)
        assert_close(g, xg.grad, 3 * rtol, 3 * atol, "t128 mulgrad")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_attention_dropout_mask(dtype):
    """dxa_attn_desc.drop_mask: torch SDPA's dropout on the attention weights (after the softmax, normaliser untouched) with the
    mask handed in, forward and backward, against autograd on the written-out formula"""
    B, H, Sq, Sk, D, p = 2, 4, 9, 21, 16, 0.1
    q, k, v = (rnd(B, H, s_, D, dtype=dtype, seed=100 + i) for i, s_ in enumerate((Sq, Sk, Sk)))
    do = rnd(B, H, Sq, D, dtype=dtype, seed=104)
    g = torch.Generator(device="cpu").manual_seed(5)
    mask = ((torch.rand(B, H, Sq, Sk, generator=g) >= p).float() / (1 - p)).to(DEV).to(dtype)
    o = torch.empty_like(q)
    lse = K.attn_fwd(q, k, v, o, causal=False, scale=D ** -0.5, drop_mask=mask)
    qr, kr, vr = (t.double().requires_grad_(True) for t in (q, k, v))
    w = torch.softmax(qr @ kr.transpose(-1, -2) * D ** -0.5, dim=-1) * mask.double()
    ref = w @ vr
    rtol, atol = (1e-5, 1e-5) if dtype == torch.float32 else (1.0 / 64, 2e-2)
    assert_close(o, ref, rtol, atol, "attention + dropout mask fwd")
    ref.backward(do.double())
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    K.attn_bwd(q, k, v, o, lse, do, dq, dk, dv, causal=False, scale=D ** -0.5, drop_mask=mask)
    for got, want, nm in ((dq, qr.grad, "dq"), (dk, kr.grad, "dk"), (dv, vr.grad, "dv")):
        assert_close(got, want, rtol * 2, atol * 2, f"attention + dropout mask {nm}")
    # without the mask the same call is the plain attention
    o2 = torch.empty_like(q)
    K.attn_fwd(q, k, v, o2, causal=False, scale=D ** -0.5, force_generic=True)
    assert_close(o2, torch.softmax(q.double() @ k.double().transpose(-1, -2) * D ** -0.5, dim=-1) @ v.double(), rtol, atol, "no mask")


def test_gemm_contraction_not_multiple_of_64_takes_the_fast_path_plus_tail():
    """K = 4304 (SigLIP-So400m's MLP width): NT with bias + residual (fc2 forward) and NN (fc1 dX) run as K0 = 4288 on the
    MFMA fast path + a 16-deep tail accumulated by the generic kernel; against fp64, and against the single generic pass
    (DXA_GEMM_NO_KTAIL semantics are exercised by the small-M call, which stays on the generic kernel)"""
    M, N, Kd = 1024, 1152, 4304
    a = rnd(M, Kd, dtype=torch.bfloat16, seed=170)
    w = rnd(N, Kd, dtype=torch.bfloat16, seed=171, scale=0.05)
    bias = rnd(N, dtype=torch.bfloat16, seed=172)
    res = rnd(M, N, dtype=torch.bfloat16, seed=173)
    ref = a.double() @ w.double().T + bias.double() + res.double()
    rtol, atol = tol_for(torch.bfloat16, Kd)
    assert_close(K.mm_nt(a, w, bias=bias, residual=res), ref, 2 * rtol, 2 * atol, "NT K=4304 bias+residual")
    small = K.mm_nt(a[:128], w, bias=bias, residual=res[:128])                     # M < 256: one generic pass
    assert_close(small, ref[:128], 2 * rtol, 2 * atol, "NT K=4304 generic")
    dy = rnd(M, Kd, dtype=torch.bfloat16, seed=174)
    w1 = rnd(Kd, N, dtype=torch.bfloat16, seed=175, scale=0.05)
    assert_close(K.mm_nn(dy, w1), dy.double() @ w1.double(), 2 * rtol, 2 * atol, "NN K=4304")
    o32 = K.mm_nt(a, w, out_dtype=torch.float32)
    assert_close(o32, a.double() @ w.double().T, 2e-5, 2e-3 * math.sqrt(Kd / 320), "NT K=4304 fp32 out")


SMALL_F32_CASES = [
    # B, Hq, Hkv, Sq, Sk, D, causal, masks
    (2, 16, 16, 17, 17, 64, False, ""),            # DiT-L self attention of a sampler step
    (2, 16, 16, 17, 256, 64, False, ""),           # MemVLA perceptual cross attention
    (3, 4, 4, 18, 18, 96, False, ""),              # DiT-S heads (384 / 4)
    (2, 12, 12, 18, 18, 64, False, "range"),
    (2, 8, 2, 33, 70, 128, True, "valid"),         # GQA, causal with Sk != Sq, key validity
    (1, 4, 1, 5, 300, 32, False, "limit,valid"),
    (2, 2, 2, 16, 1, 64, False, ""),               # one key
    (1, 2, 2, 20, 2048, 64, True, "range"),        # the largest score slab (16 x 2052 floats of LDS)
    (2, 3, 3, 40, 40, 64, True, "allmasked"),      # a batch row with no visible key at all
]


@pytest.mark.parametrize("case", SMALL_F32_CASES)
def test_attention_small_fp32_forward(case):
    """attn_fwd_small_f32_k (fp32 heads of the diffusion samplers: 16 queries of one head per workgroup, exact fp32 MFMA for
    Q K^T and P V, softmax in LDS) against fp64 and against the one-wave-per-row kernel, over every mask the descriptor has,
    token-major (fused qkv) strides, GQA, and the head sizes it is instantiated for"""
    B, Hq, Hkv, Sq, Sk, D, causal, masks = case
    scale = D ** -0.5
    if Sq == Sk and Hq == Hkv:
        qkv = rnd(B, Sq, 3, Hq, D, dtype=torch.float32, seed=170)          # token-major fused projection output
        q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    else:
        q = rnd(B, Hq, Sq, D, dtype=torch.float32, seed=171)
        kv = rnd(B, Sk, 2, Hkv, D, dtype=torch.float32, seed=172)           # [k | v] of a packed in-projection
        k, v = kv[:, :, 0].permute(0, 2, 1, 3), kv[:, :, 1].permute(0, 2, 1, 3)
    o_store = torch.empty(B, Sq, Hq, D, device=DEV, dtype=torch.float32)
    o = o_store.permute(0, 2, 1, 3)
    o1 = torch.empty(B, Hq, Sq, D, device=DEV, dtype=torch.float32)
    kw = dict(causal=causal, scale=scale)
    vis = torch.ones(B, Sq, Sk, dtype=torch.bool, device=DEV)
    j = torch.arange(Sk, device=DEV)[None, None, :]
    i = torch.arange(Sq, device=DEV)[None, :, None]
    if "range" in masks:
        kw["kv_start"] = torch.tensor([b % 3 for b in range(B)], dtype=torch.int32, device=DEV)
        kw["kv_end"] = torch.tensor([Sk - 2 * b for b in range(B)], dtype=torch.int32, device=DEV)
        vis &= (j >= kw["kv_start"].view(B, 1, 1)) & (j < kw["kv_end"].view(B, 1, 1))
    if "allmasked" in masks:
        kw["kv_end"] = torch.tensor([Sk, 0][:B], dtype=torch.int32, device=DEV)
        vis &= j < kw["kv_end"].view(B, 1, 1)
    if "valid" in masks:
        g = torch.Generator(device="cpu").manual_seed(9)
        valid = (torch.rand(B, Sk, generator=g) > 0.3).to(DEV)
        kw["key_valid"] = valid.to(torch.uint8)
        vis &= valid[:, None, :]
    if "limit" in masks:
        kw["q_limit"] = torch.tensor([[Sk - 7 * (r % 4) for r in range(Sq)]] * B, dtype=torch.int32, device=DEV)
        vis &= j < kw["q_limit"][:, :, None]
    if causal:
        vis &= j <= i + (Sk - Sq)
    lse = K.attn_fwd(q, k, v, o, **kw)
    lse1 = K.attn_fwd(q, k, v, o1, force_generic=True, **kw)
    G = Hq // Hkv
    sc = q.double() @ k.double().repeat_interleave(G, 1).transpose(-1, -2) * scale
    sc = sc.masked_fill(~vis[:, None], float("-inf"))
    ref = torch.nan_to_num(torch.softmax(sc, -1), nan=0.0) @ v.double().repeat_interleave(G, 1)
    assert_close(o, ref, 2e-5, 2e-5, "small fp32 fwd")
    assert_close(o, o1.double(), 2e-5, 2e-5, "small vs one-wave-per-row")
    live = vis.any(-1)[:, None, :].expand(B, Hq, Sq)
    assert_close(lse[live], torch.logsumexp(sc, -1)[live], 1e-4, 1e-4, "small fp32 lse")
    assert bool(torch.all(lse[~live] == 0))                                # rows with no visible key: o = 0, lse = 0
    assert_close(lse, lse1.double(), 1e-5, 1e-5, "small vs one-wave-per-row lse")
    o2 = torch.empty_like(o1)
    K.attn_fwd(q, k, v, o2, **kw)
    assert torch.equal(o2, o.contiguous())                                 # run to run bit-identical


@pytest.mark.parametrize("case", [(1, 8, 1, 17, 833, 256), (2, 8, 1, 17, 833, 256), (1, 28, 4, 1, 700, 128), (3, 4, 4, 40, 1000, 64),
                                  (1, 8, 1, 17, 300, 256)])
def test_attention_flash_key_range_split(case):
    """few queries against a long key cache (pi0's KV-cached denoising step: 17 suffix queries x 8 heads over prefix + suffix
    keys; a decode step): the flash forward runs one workgroup per KEY RANGE and attn_split_combine_k folds the normalised
    partials by their log-sum-exps.  Block-prefix limits, key validity, a batch row whose cache ends early (whole ranges empty)
    and — in the last batch row of the multi-row cases — no visible key at all; o and lse against fp64, repetitions bitwise"""
    B, Hq, Hkv, Sq, Sk, D = case
    dtype = torch.bfloat16
    q, k, v = rnd(B, Hq, Sq, D, dtype=dtype, seed=210), rnd(B, Hkv, Sk, D, dtype=dtype, seed=211), rnd(B, Hkv, Sk, D, dtype=dtype, seed=212)
    g = torch.Generator(device="cpu").manual_seed(12)
    valid = (torch.rand(B, Sk, generator=g) > 0.1).to(DEV)
    kv_end = torch.tensor([Sk if b == 0 else (130 if b == 1 else Sk) for b in range(B)], dtype=torch.int32, device=DEV)
    if B >= 3:
        valid[B - 1] = False                                               # nothing visible: o = 0, lse = 0
    q_limit = torch.full((B, Sq), Sk, dtype=torch.int32, device=DEV)
    q_limit[:, 0] = Sk - Sq + 1                                            # the first query does not see the other new keys
    scale = D ** -0.5
    o = torch.empty(B, Sq, Hq, D, device=DEV, dtype=dtype).permute(0, 2, 1, 3)
    kw = dict(causal=False, scale=scale, kv_end=kv_end, q_limit=q_limit, key_valid=valid.to(torch.uint8))
    lse = K.attn_fwd(q, k, v, o, **kw)
    j = torch.arange(Sk, device=DEV)[None, None, :]
    vis = (j < kv_end.view(B, 1, 1)) & (j < q_limit[:, :, None]) & valid[:, None, :]
    G = Hq // Hkv
    sc = q.double() @ k.double().repeat_interleave(G, 1).transpose(-1, -2) * scale
    sc = sc.masked_fill(~vis[:, None], float("-inf"))
    ref = torch.nan_to_num(torch.softmax(sc, -1), nan=0.0) @ v.double().repeat_interleave(G, 1)
    assert_close(o, ref, 1.0 / 64, 2e-2, "split-range flash o")
    live = vis.any(-1)[:, None, :].expand(B, Hq, Sq)
    assert_close(lse[live], torch.logsumexp(sc, -1)[live], 1e-2, 3e-2, "split-range flash lse")
    assert bool(torch.all(lse[~live] == 0)) and bool(torch.all(o.float()[~live] == 0))
    o2 = torch.empty_like(o)
    lse2 = K.attn_fwd(q, k, v, o2, **kw)
    assert torch.equal(o2, o) and torch.equal(lse2, lse)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_attention_materialised_forward_masks_dropout_and_sizes(dtype):
    """dxa_attn_fwd_ws: the eager-style forward (S = Q K^T by the batched GEMM, masked row softmax, O = P V) that large non-flash
    problems take — pi0 served in fp32 (MQA, head_dim 256, suffix queries against prefix + suffix keys, block-prefix limits, key
    validity) and retrieval attention with a dropout mask — against the one-wave-per-row kernel and an explicit reference"""
    B, Hq, Hkv, Sq, Sk, D = 2, 8, 1, 51, 300, 256
    q, k, v = rnd(B, Hq, Sq, D, dtype=dtype, seed=150), rnd(B, Hkv, Sk, D, dtype=dtype, seed=151), rnd(B, Hkv, Sk, D, dtype=dtype, seed=152)
    valid = torch.ones(B, Sk, dtype=torch.bool, device=DEV)
    valid[0, 40:60] = False
    q_limit = torch.full((B, Sq), Sk, dtype=torch.int32, device=DEV)
    q_limit[:, 0] = Sk - 50                                           # the state token does not see the action tokens
    scale = D ** -0.5
    o, o1 = torch.empty_like(q), torch.empty_like(q)
    kw = dict(causal=False, scale=scale, q_limit=q_limit, key_valid=valid.to(torch.uint8))
    lse = K.attn_fwd(q, k, v, o, **kw)                                # Sq >= 32, Sk >= 128, not flash-eligible in fp32: materialised
    lse1 = K.attn_fwd(q, k, v, o1, force_generic=True, **kw)
    mask = (torch.arange(Sk, device=DEV)[None, None, :] < q_limit[:, :, None]) & valid[:, None, :]
    sc = torch.einsum("bhid,bjd->bhij", q.double(), k.double()[:, 0]) * scale
    sc = sc.masked_fill(~mask[:, None], float("-inf"))
    ref = torch.softmax(sc, -1) @ v.double()[:, :1]
    rtol, atol = (2e-5, 2e-5) if dtype == torch.float32 else (1.0 / 64, 2e-2)
    if dtype == torch.float32:
        assert_close(o, ref, rtol, atol, "materialised fwd")
        assert_close(lse, torch.logsumexp(sc, -1), 1e-4, 1e-4, "materialised lse")
    assert_close(o, o1.double(), rtol, atol, "materialised vs one-wave-per-row")
    assert_close(lse, lse1.double(), 1e-4 if dtype == torch.float32 else 1e-2, 1e-4 if dtype == torch.float32 else 3e-2, "lse")
    # dropout mask, 4-head retrieval shape (256 queries x 1024 keys, head_dim 64)
    B, H, Sq, Sk, D = 1, 4, 256, 1024, 64
    q, k, v = (rnd(B, H, s_, D, dtype=dtype, seed=160 + i) for i, s_ in enumerate((Sq, Sk, Sk)))
    g = torch.Generator(device="cpu").manual_seed(6)
    dm = ((torch.rand(B, H, Sq, Sk, generator=g) >= 0.1).float() / 0.9).to(DEV).to(dtype)
    o = torch.empty_like(q)
    K.attn_fwd(q, k, v, o, causal=False, scale=D ** -0.5, drop_mask=dm)
    ref = (torch.softmax(q.double() @ k.double().transpose(-1, -2) * D ** -0.5, -1) * dm.double()) @ v.double()
    assert_close(o, ref, rtol, atol, "materialised fwd + dropout")


@pytest.mark.parametrize("M,Kd,N", [(1, 14336, 3584), (1, 3584, 3584), (3, 3584, 14336), (8, 200, 520), (5, 64, 64), (2, 1030, 72)])
def test_gemv_few_row_nn(M, Kd, N):
    """dX of a linear applied to a handful of tokens (MemVLA's one-token cognition stream): out[M <= 8, N] = dy[M, K] W[K, N]
    as a K-sliced stream over W as it lies; against fp64, bf16 and fp32 outputs, repetitions bitwise equal"""
    dy = rnd(M, Kd, dtype=torch.bfloat16, seed=111)
    w = rnd(Kd, N, dtype=torch.bfloat16, seed=112, scale=0.1)
    ref = dy.double() @ w.double()
    rtol, atol = tol_for(torch.bfloat16, Kd)
    o1 = K.mm_nn(dy, w)
    assert_close(o1, ref, 2 * rtol, 2 * atol, "few-row nn bf16")
    o32 = K.mm_nn(dy, w, out_dtype=torch.float32)
    assert_close(o32, ref, 2e-5, 2e-3 * math.sqrt(max(Kd, 320) / 320), "few-row nn f32")
    assert torch.equal(o1, K.mm_nn(dy, w)) and torch.equal(o32, K.mm_nn(dy, w, out_dtype=torch.float32))


@pytest.mark.parametrize("M,N,K1,K2", [(512, 384, 1000, 1000), (300, 520, 200, 77), (3584, 1024, 2296, 2296), (96, 96, 130, 64)])
@pytest.mark.parametrize("accum", [False, True])
def test_tn_two_segments_is_one_product_over_both(M, N, K1, K2, accum):
    """dxa_gemm_desc.A2 / B2 / K2: C = A^T B + A2^T B2 in one pass over C — the weight gradient over two micro-batches
    (ping-pong kernel: both segments inside the K loop, segment 1's ragged last tile zero-filled by its own descriptor;
    small / odd shapes: two products, the second accumulating).  Held to the fp64 product over the concatenated rows, with the
    mirror and the sum of squares of the final values."""
    g = torch.Generator(device=DEV).manual_seed(M + N + K1)
    r = lambda *s: (torch.rand(*s, device=DEV, generator=g) * 2 - 1).bfloat16()
    a, b, a2, b2 = r(K1, M), r(K1, N), r(K2, M), r(K2, N)
    c0 = torch.rand(M, N, device=DEV, generator=g)
    out = c0.clone() if accum else torch.empty(M, N, device=DEV)
    mir = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    part = torch.full((K.gemm_sumsq_slots(M, N),), float("nan"), device=DEV)
    K.mm_tn(a, b, out=out, accumulate=accum, mirror=mir, sumsq=part, a2=a2, b2=b2)
    ref = torch.cat([a, a2]).double().t() @ torch.cat([b, b2]).double() + (c0.double() if accum else 0)
    assert ((out.double() - ref).abs().max() / ref.abs().max()).item() < 2e-6
    assert ((mir.double() - ref).abs().max() / ref.abs().max()).item() < 8e-3
    want = (out.double() ** 2).sum().item()
    assert abs(part.double().sum().item() - want) < 1e-5 * want
    # and it equals the two-pass form bit for bit where both run the same kernel in the same K order is NOT promised: the
    # one-pass form keeps a single fp32 accumulation chain; the bound above is what both satisfy
    with pytest.raises(L.DxaError):
        K.mm_nt(r(64, 64), r(64, 64), a2=a2, b2=b2)


@pytest.mark.parametrize("M,F_,Kd,split_tail", [(300, 512, 256, False), (600, 1024, 512, False), (543, 520, 192, False),
                                                (1100, 2056, 320, False), (543, 18944, 3584, False), (4592, 18944, 3584, True)])
def test_gemm_swiglu_epilogue_is_bit_identical_to_gemm_then_swiglu(M, F_, Kd, split_tail):
    """dxa_gemm_desc.fuse = DXA_FUSE_SWIGLU (gate / up product with silu(gate) * up in the epilogue, 256- and 192-row tiles, ragged
    edges, the decoder's real shapes) against mm_nt + swiglu_fwd: the output is swiglu_fwd of the stored pre-activations bit for bit,
    and the pre-activations are those of the plain product bit for bit — except where a split-K tail exists (the training shape:
    2664 tiles = 10 rounds of 256 + 104 tail tiles cut along K): the fused tile (128 gate + 128 up columns) and the plain tile (256
    consecutive columns) put different outputs into the tail tiles, whose fp32 partial sums are added in another order, so a few
    values land on the neighbouring bf16"""
    g = torch.Generator(device="cpu").manual_seed(M + F_)
    x = (torch.randn(M, Kd, generator=g) * 0.7).to(DEV).to(torch.bfloat16)
    w = (torch.randn(2 * F_, Kd, generator=g) * (1.5 / math.sqrt(Kd))).to(DEV).to(torch.bfloat16)
    assert K.swiglu_gemm_supported(x, w, keep_pre=False)
    pre_ref = K.mm_nt(x, w)
    out_ref = K.swiglu_fwd(pre_ref)
    out, pre = K.mm_nt_swiglu(x, w, keep_pre=True)
    out2, none = K.mm_nt_swiglu(x, w, keep_pre=False)
    torch.cuda.synchronize()
    assert none is None
    mine = K.swiglu_fwd(pre)                                          # the epilogue is swiglu_fwd up to its fast exp / reciprocal
    off = (out.view(torch.int16) != mine.view(torch.int16)).float().mean().item()
    step = ((out.float() - mine.float()).abs() / mine.float().abs().clamp_min(1e-6)).max().item()
    print(f"M {M} F {F_} K {Kd}: {off:.2e} of the outputs differ from swiglu_fwd of the stored pre-activations, by at most {step:.2e}")
    assert off < 1e-3 and step <= 2 ** -7 * 1.01
    assert torch.equal(out2.view(torch.int16), out.view(torch.int16))
    if not split_tail:                                                # no tile is cut along K: same summation order everywhere
        assert torch.equal(pre.view(torch.int16), pre_ref.view(torch.int16))
    else:
        differ = (pre.view(torch.int16) != pre_ref.view(torch.int16)).float().mean().item()
        worst = ((pre.float() - pre_ref.float()).abs() / pre_ref.float().abs().clamp_min(1e-3)).max().item()
        print(f"M {M} F {F_} K {Kd}: {differ:.2e} of the pre-activations differ, by at most {worst:.2e} relative")
        assert differ < 2e-2 and worst <= 2 ** -7 * 1.01                                    # one bf16 step
    assert out_ref.float().abs().max() > 0.05                         # (not a comparison of zeros)


@pytest.mark.parametrize("Sq,Sk", [(17, 17), (17, 256), (32, 70), (5, 33), (68, 256), (96, 40), (33, 17), (65, 31)])
def test_attention_small_f32_backward_one_launch(Sq, Sk, monkeypatch):
    """the head-sized fp32 attention backward (attn_bwd_small_f32_k: the DiT heads' 17 x 17 and MemVLA's 17 x 256 perceptual attention,
    one launch, queries and keys in chunks of 32; 68 = the 4 x 17 queries of a MemVLA sample's diffusion repeats) against double-precision autograd and against the generic eight-launch path"""
    B, H, D = 3, 4, 64
    scale = D ** -0.5
    q, k, v = rnd(B, H, Sq, D, dtype=torch.float32, seed=90), rnd(B, H, Sk, D, dtype=torch.float32, seed=91), rnd(B, H, Sk, D, dtype=torch.float32, seed=92)
    do = rnd(B, H, Sq, D, dtype=torch.float32, seed=93)
    o = torch.empty(B, H, Sq, D, device=DEV, dtype=torch.float32)
    lse = K.attn_fwd(q, k, v, o, causal=False, scale=scale)
    qr, kr, vr = (t.double().detach().clone().requires_grad_(True) for t in (q, k, v))
    pr = torch.softmax(qr @ kr.transpose(-1, -2) * scale, -1)
    (pr @ vr).backward(do.double())
    outs = {}
    for generic in (False, True):
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        K.attn_bwd(q, k, v, o, lse, do, dq, dk, dv, causal=False, scale=scale, force_generic=generic)
        assert_close(dq, qr.grad, 1e-4, 1e-4, "dq")
        assert_close(dk, kr.grad, 1e-4, 2e-4, "dk")
        assert_close(dv, vr.grad, 1e-4, 2e-4, "dv")
        outs[generic] = (dq, dk, dv)
    for a, b_ in zip(outs[False], outs[True]):
        assert float((a - b_).abs().max()) <= 2e-5 * float(b_.abs().max()) + 1e-6


def test_split3_pair_writes_what_the_single_launches_write():
    """dxa_split3_pair: both operand splits of a bf16x3 product in one launch — plain / transposed in every combination, ragged
    extents, a padded transposed operand, an explicit leading dimension — bit for bit what dxa_split3 / dxa_split3_t write"""
    a = rnd(1088, 768, dtype=torch.float32, seed=70)
    w = rnd(2304, 768, dtype=torch.float32, seed=71)
    dy = rnd(1088, 2304, dtype=torch.float32, seed=72)
    wide = rnd(300, 520, dtype=torch.float32, seed=73)
    a3, w3 = K.split3_pair(a, False, w, False)
    assert torch.equal(a3, K.split3(a, 1088, 768, 768, 0)) and torch.equal(w3, K.split3(w, 2304, 768, 768, 1))
    d3, wt3 = K.split3_pair(dy, False, w, True)                                    # dX = dY W as an NT product against W^T
    assert torch.equal(d3, K.split3(dy, 1088, 2304, 2304, 0)) and torch.equal(wt3, K.split3_t(w, 1, 1))
    dt3, at3 = K.split3_pair(dy, True, a, True, pad_to=32)                        # dW = dY^T X
    assert torch.equal(dt3, K.split3_t(dy, 32, 0)) and torch.equal(at3, K.split3_t(a, 32, 1))
    assert dt3.shape == (2304, 3 * 1088)
    r3, s3 = K.split3_pair(wide, True, wide, False, pad_to=32, b_dims=(200, 256, 520))
    assert r3.shape == (520, 3 * 320) and torch.equal(r3, K.split3_t(wide, 32, 0))
    assert torch.equal(s3, K.split3(wide, 200, 256, 520, 1))
    e3, f3 = K.split3_pair(a[:0], False, w, False)                                 # an empty operand is skipped, the other one written
    assert e3.numel() == 0 and torch.equal(f3, w3)


@pytest.mark.parametrize("Sq,Sk,D", [(68, 256, 32), (40, 100, 48), (96, 129, 64)])
def test_attention_small_f32_backward_register_blocked_token_major(Sq, Sk, D):
    """more than 32 queries per (sample, head): attn_bwd_small2_f32_k (every thread a block of outputs in registers, 16-byte LDS reads;
    MemVLA's 68 x 256 perceptual attention) on token-major [B, S, H, D] operands as the DiT blocks hold them, head widths below 64,
    a ragged last key chunk — against double-precision autograd and the generic path"""
    B, H = 2, 3
    scale = D ** -0.5
    mk = lambda S, seed: rnd(B, S, H, D, dtype=torch.float32, seed=seed).permute(0, 2, 1, 3)
    q, k, v, do = mk(Sq, 80), mk(Sk, 81), mk(Sk, 82), mk(Sq, 83)
    o = torch.empty(B, Sq, H, D, device=DEV, dtype=torch.float32).permute(0, 2, 1, 3)
    lse = K.attn_fwd(q, k, v, o, causal=False, scale=scale)
    qr, kr, vr = (t.double().detach().clone().requires_grad_(True) for t in (q, k, v))
    (torch.softmax(qr @ kr.transpose(-1, -2) * scale, -1) @ vr).backward(do.double())
    outs = {}
    for generic in (False, True):
        dq, dk, dv = (torch.full((B, S, H, D), float("nan"), device=DEV).permute(0, 2, 1, 3) for S in (Sq, Sk, Sk))
        K.attn_bwd(q, k, v, o, lse, do, dq, dk, dv, causal=False, scale=scale, force_generic=generic)
        assert_close(dq, qr.grad, 1e-4, 1e-4, "dq")
        assert_close(dk, kr.grad, 1e-4, 2e-4, "dk")
        assert_close(dv, vr.grad, 1e-4, 2e-4, "dv")
        outs[generic] = (dq, dk, dv)
    for a, b_ in zip(outs[False], outs[True]):
        assert float((a - b_).abs().max()) <= 2e-5 * float(b_.abs().max()) + 1e-6
