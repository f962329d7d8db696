"""Tensor-level wrappers over the C ABI (no autograd here — see functional.py).

Every function takes torch tensors living on the MI355X, passes raw device pointers + sizes + the
current HIP stream to libdexbotic_amd.so and returns torch tensors allocated by torch's caching
allocator (PyTorch = device memory and streams only).  Nothing here computes with torch ops.
"""
from __future__ import annotations

import contextlib
import ctypes as C
from typing import Optional, Sequence, Tuple

import torch

from . import _lib as L
from ._lib import lib

F32, BF16 = L.F32, L.BF16


def dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise L.DxaError(f"unsupported dtype {t.dtype} (fp32 / bf16 only)")


def torch_dtype(code: int) -> torch.dtype:
    return torch.bfloat16 if code == BF16 else torch.float32


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream() -> int:
    """raw handle of the calling thread's current stream on the current device (what every launch below is enqueued on).  The two
    torch._C calls cost ~0.3 us against ~1.5 us for torch.cuda.current_stream().cuda_stream (a Stream object per call) — on a step of
    2,000 - 7,700 launches that is host time the launch-bound stretches wait for"""
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise L.DxaError("dexbotic_amd kernels need device tensors (no CPU fallback exists)")
    return t.data_ptr()


def _row_major(t: torch.Tensor, name: str) -> int:
    """leading dimension of a 2-D tensor whose last dim is contiguous"""
    if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise L.DxaError(f"{name}: expected a 2-D tensor with contiguous rows, got shape {tuple(t.shape)} "
                         f"strides {t.stride()}")
    return t.stride(0) if t.shape[0] > 1 else max(t.shape[1], t.stride(0))


# ------------------------------------------------------------------------------------------------ GEMM
# How fp32 x fp32 NT products of >= 128 rows are computed: "exact" = fp32 MFMA (v_mfma_f32_16x16x4_f32, 157 TF/s peak);
# "bf16x3" = three-term split-bf16 product on the bf16 ring kernel (16 mantissa bits per operand, ~1e-5 relative; the
# reference's trainer runs the same fp32 head matmuls as TF32, base_exp.py:254).  Set by NativeTrainer.step /
# inference_action from the model config; parity tests in fp32 mode leave it at "exact".
F32_GEMM_MODE = "exact"


@contextlib.contextmanager
def f32_gemm_mode(mode: str):
    global F32_GEMM_MODE
    if mode not in ("exact", "bf16x3"):
        raise L.DxaError(f"unknown fp32 GEMM mode {mode!r}")
    prev, F32_GEMM_MODE = F32_GEMM_MODE, mode
    try:
        yield
    finally:
        F32_GEMM_MODE = prev


def split3(x: torch.Tensor, rows: int, cols: int, ld: int, side: int) -> torch.Tensor:
    """fp32 [rows, cols] -> bf16 [rows, 3*cols] = [hi | hi | lo] (side 0) or [hi | lo | hi] (side 1), x = hi + lo"""
    out = torch.empty((rows, 3 * cols), device=x.device, dtype=torch.bfloat16)
    L.check(lib.dxa_split3(_ptr(x), ld, _ptr(out), rows, cols, side, _stream()), "dxa_split3")
    return out


def split3_t(x: torch.Tensor, pad_to: int, side: int) -> torch.Tensor:
    """fp32 [R, C] -> bf16 [C, 3 * Rp] = split3 of x^T (Rp = R rounded up to ``pad_to``, zero filled) in ONE launch"""
    R, C_ = x.shape
    Rp = (R + pad_to - 1) // pad_to * pad_to
    out = torch.empty((C_, 3 * Rp), device=x.device, dtype=torch.bfloat16)
    L.check(lib.dxa_split3_t(_ptr(x), _row_major(x, "x"), _ptr(out), R, C_, Rp, side, _stream()), "dxa_split3_t")
    return out


_NO_SPLIT_PAIR = __import__("os").environ.get("DXA_NO_SPLIT_PAIR") == "1"


def split3_pair(a: torch.Tensor, a_t: bool, b: torch.Tensor, b_t: bool, pad_to: int = 1, a_dims=None, b_dims=None
                ) -> Tuple[torch.Tensor, torch.Tensor]:
    """both operands of one bf16x3 product in ONE launch (dxa_split3_pair): ``a`` as the A operand (hi | hi | lo), ``b`` as the B operand
    (hi | lo | hi); ``*_t``: the split of the TRANSPOSE ([R, C] -> [C, 3 Rp], Rp = R rounded up to ``pad_to``) as split3_t writes it;
    ``*_dims`` = (rows, cols, ld) of an operand that is not simply a row-major 2-D tensor (gemm()'s explicit extents)"""
    if _NO_SPLIT_PAIR:                                 # A/B: one launch per operand (rounds 2 - 6b)
        res = []
        for x, t, side, dims in ((a, a_t, 0, a_dims), (b, b_t, 1, b_dims)):
            res.append(split3_t(x, pad_to, side) if t else split3(x, *(dims if dims is not None else (x.shape[0], x.shape[1], _row_major(x, "x"))), side))
        return res[0], res[1]
    ops, outs = [], []
    for x, t, side, dims in ((a, a_t, 0, a_dims), (b, b_t, 1, b_dims)):
        R, C_, ld = dims if dims is not None else (x.shape[0], x.shape[1], _row_major(x, "x"))
        o = L.Split3Op()
        o.src, o.ld, o.rows, o.cols, o.side, o.transposed = _ptr(x), ld, R, C_, side, int(t)
        if t:
            Rp = (R + pad_to - 1) // pad_to * pad_to
            out = torch.empty((C_, 3 * Rp), device=x.device, dtype=torch.bfloat16)
            o.pad = Rp
        else:
            out = torch.empty((R, 3 * C_), device=x.device, dtype=torch.bfloat16)
            o.pad = 0
        o.dst = _ptr(out)
        ops.append(o)
        outs.append(out)
    L.check(lib.dxa_split3_pair(C.byref(ops[0]), C.byref(ops[1]), _stream()), "dxa_split3_pair")
    return outs[0], outs[1]


def _x3_t_ok(M: int, N: int, Kc: int, out: torch.Tensor) -> bool:
    """the split-bf16 NT product is admissible for these extents (same rule as _x3_eligible; Kc = contraction length)"""
    return (F32_GEMM_MODE == "bf16x3" and out.dtype == torch.float32 and M >= 128 and N >= 128 and Kc >= 64 and Kc % 32 == 0
            and 6 * max(M, N) * Kc < (1 << 31))


def mm_tn_f32(a: torch.Tensor, b: torch.Tensor, *, out: torch.Tensor, **kw) -> torch.Tensor:
    """out[M,N] = a[K,M]^T @ b[K,N] for fp32 operands as an NT product of the transposes (the fp32 heads' dW).  bf16x3 mode: the
    transposed split operands come out of ONE launch each (dxa_split3_t); exact mode: fp32 transposes + the fp32 MFMA kernel."""
    Kc, M = a.shape
    N = b.shape[1]
    if a.dtype == torch.float32 and b.dtype == torch.float32 and _x3_t_ok(M, N, (Kc + 31) // 32 * 32, out):
        a3, b3 = split3_pair(a, True, b, True, pad_to=32)
        Kp = a3.shape[1] // 3
        kw = _ld_kwargs(kw, out)
        return gemm(L.NT, a3, b3, M, N, 3 * Kp, 3 * Kp, 3 * Kp, out, _row_major(out, "out"), epi_f32=True, **kw)
    return mm_nt(transpose(a, 4), transpose(b, 4), out=out, **kw)


def mm_nn_f32(a: torch.Tensor, b: torch.Tensor, **kw) -> torch.Tensor:
    """a[M,K] @ b[K,N] for fp32 operands as an NT product against b^T (the fp32 heads' dX = dY W)"""
    M, Kc = a.shape
    N = b.shape[1]
    out = kw.get("out")
    if (a.dtype == torch.float32 and b.dtype == torch.float32 and Kc % 32 == 0 and a.is_contiguous() and a.data_ptr() % 16 == 0
            and Kc % 4 == 0 and (out is None or out.dtype == torch.float32)):
        if out is None:
            out = kw["out"] = torch.empty((M, N), device=a.device, dtype=torch.float32)
        if _x3_t_ok(M, N, Kc, out):
            a3, b3 = split3_pair(a, False, b, True)
            kw2 = _ld_kwargs({k: v for k, v in kw.items() if k != "out"}, out)
            return gemm(L.NT, a3, b3, M, N, 3 * Kc, 3 * Kc, 3 * Kc, out, _row_major(out, "out"), epi_f32=True, **kw2)
    return mm_nt(a, transpose(b, 1), **kw)


def _x3_eligible(layout, a, b, out, M, N, K, lda, ldb, nb) -> bool:
    return (F32_GEMM_MODE == "bf16x3" and layout == L.NT and a.dtype == torch.float32 and b.dtype == torch.float32
            and out.dtype == torch.float32 and tuple(nb) == (1, 1, 1) and M >= 128 and N >= 128 and K >= 64 and K % 32 == 0
            and lda % 4 == 0 and ldb % 4 == 0 and a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0
            and 6 * max(M, N) * K < (1 << 31))


_NB1 = (1, 1, 1)


def gemm(layout: int, a: torch.Tensor, b: torch.Tensor, M: int, N: int, K: int, lda: int, ldb: int,
         out: torch.Tensor, ldc: int, *, bias: Optional[torch.Tensor] = None,
         residual: Optional[torch.Tensor] = None, ldr: int = 0, act: int = L.ACT_NONE,
         aux_out: Optional[torch.Tensor] = None, mulgrad: Optional[torch.Tensor] = None, ldg: int = 0,
         alpha: float = 1.0, accumulate: bool = False, nb: Sequence[int] = _NB1,
         sA: Sequence[int] = (0, 0, 0), sB: Sequence[int] = (0, 0, 0), sC: Sequence[int] = (0, 0, 0),
         sR: Sequence[int] = (0, 0, 0), sG: Sequence[int] = (0, 0, 0), epi_f32: bool = False,
         mirror: Optional[torch.Tensor] = None, sumsq: Optional[torch.Tensor] = None,
         a2: Optional[torch.Tensor] = None, b2: Optional[torch.Tensor] = None) -> torch.Tensor:
    if not epi_f32 and a2 is None and _x3_eligible(layout, a, b, out, M, N, K, lda, ldb, nb):
        a3, b3 = split3_pair(a, False, b, False, a_dims=(M, K, lda), b_dims=(N, K, ldb))
        return gemm(L.NT, a3, b3, M, N, 3 * K, 3 * K, 3 * K, out, ldc, bias=bias, residual=residual, ldr=ldr, act=act,
                    aux_out=aux_out, mulgrad=mulgrad, ldg=ldg, alpha=alpha, accumulate=accumulate, epi_f32=True,
                    mirror=mirror, sumsq=sumsq)
    d = L.GemmDesc()
    d.layout, d.in_dtype, d.out_dtype, d.act = layout, dt(a), dt(out), act
    d.epi_f32 = int(epi_f32)
    if dt(b) != d.in_dtype:
        raise L.DxaError("gemm: A and B dtypes differ")
    for t, n in ((bias, "bias"), (residual, "residual"), (mulgrad, "mulgrad")):
        if t is not None and dt(t) != (F32 if epi_f32 else d.in_dtype):
            raise L.DxaError(f"gemm: {n} dtype must equal the input dtype")
    if aux_out is not None and dt(aux_out) != d.out_dtype:
        raise L.DxaError("gemm: aux_out dtype must equal the output dtype")
    d.M, d.N, d.K = M, N, K
    d.A, d.lda, d.B, d.ldb, d.C, d.ldc = _ptr(a), lda, _ptr(b), ldb, _ptr(out), ldc
    d.bias, d.residual, d.ldr = _ptr(bias), _ptr(residual), ldr
    d.aux_out, d.mulgrad, d.ldg = _ptr(aux_out), _ptr(mulgrad), ldg
    d.alpha, d.accumulate = alpha, int(accumulate)
    if mirror is not None:
        if mirror.dtype != torch.bfloat16 or out.dtype != torch.float32 or tuple(nb) != (1, 1, 1) or \
                _row_major(mirror, "mirror") != ldc:
            raise L.DxaError("gemm: mirror must be a bf16 [M, N] view with the output's leading dimension (fp32 output)")
        d.mirror = _ptr(mirror)
    if sumsq is not None:
        if sumsq.dtype != torch.float32 or tuple(nb) != (1, 1, 1) or not sumsq.is_contiguous() \
                or sumsq.numel() != gemm_sumsq_slots(M, N):
            raise L.DxaError("gemm: sumsq must be gemm_sumsq_slots(M, N) contiguous floats (unbatched output)")
        d.sumsq = _ptr(sumsq)
    if nb is _NB1 or tuple(nb) == (1, 1, 1):         # (a fresh descriptor is zeroed: the strides of an unbatched product stay 0)
        d.nb[0] = d.nb[1] = d.nb[2] = 1
    else:
        for i in range(3):
            d.nb[i], d.sA[i], d.sB[i], d.sC[i], d.sR[i], d.sG[i] = nb[i], sA[i], sB[i], sC[i], sR[i], sG[i]
    K_all = K
    if a2 is not None:                         # TN: a second pair of operands contracted into the same product
        if layout != L.TN or b2 is None or a2.shape[1] != M or b2.shape[1] != N or a2.shape[0] != b2.shape[0] \
                or dt(a2) != d.in_dtype or dt(b2) != d.in_dtype or _row_major(a2, "a2") != lda or _row_major(b2, "b2") != ldb:
            raise L.DxaError("gemm: a2 / b2 are a second [K2, M] / [K2, N] pair of a TN product (same dtypes and leading dimensions)")
        d.A2, d.B2, d.K2 = _ptr(a2), _ptr(b2), a2.shape[0]
        K_all = K + a2.shape[0]
    prof = GEMM_PROFILE
    if prof is not None and prof.wants(layout, d.in_dtype, d.out_dtype):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.check(lib.dxa_gemm(C.byref(d), _stream()), "dxa_gemm")
        e1.record()
        batches = nb[0] * nb[1] * nb[2]
        esz = {L.BF16: 2, L.F32: 4}
        prof.add((layout, d.in_dtype, d.out_dtype), e0, e1, 2.0 * M * N * K_all * batches,
                 float((M * K_all + N * K_all) * esz.get(d.in_dtype, 2) + M * N * esz.get(d.out_dtype, 2)) * batches)
        return out
    L.check(lib.dxa_gemm(C.byref(d), _stream()), "dxa_gemm")
    return out


class GemmProfile:
    """HIP-event timing of launches of the chosen gemm instantiations — keys (layout, in dtype, out dtype) — on the
    stream they are launched on: bench.py's live roofline measurement.  Per timed launch it also records the algorithmic
    flops (2 M N K) and the algorithmic bytes (A and B read once, C written once).  ``stride`` = s: every s-th launch of a key
    is timed (a deterministic sample spread over the whole region; s coprime with the 4 products a transformer layer launches
    per layout, so every product is sampled equally often) — a pair of event records costs the stream ~4 us, and 1240 pairs per
    step were 2 % of the step they were there to measure.  ``seen`` counts every launch of a key, timed or not."""

    def __init__(self, *keys, stride: int = 1):
        if len(keys) == 3 and all(isinstance(k, int) for k in keys):
            keys = (tuple(keys),)
        self.keys = set(tuple(k) for k in keys)
        self.events = []
        self.stride = max(1, int(stride))
        self.seen = {k: 0 for k in self.keys}

    def wants(self, layout, in_dtype, out_dtype) -> bool:
        key = (layout, in_dtype, out_dtype)
        if key not in self.keys:
            return False
        n = self.seen[key]
        self.seen[key] = n + 1
        return n % self.stride == 0

    def add(self, key, e0, e1, flops: float, nbytes: float = 0.0) -> None:
        self.events.append((key, e0, e1, flops, nbytes))

    def summary(self, key=None):
        """(timed launches, their total_ms, total_flops, total_algorithmic_bytes), of one key or of all — call after a device sync"""
        ev = [e for e in self.events if key is None or e[0] == tuple(key)]
        ms = sum(a.elapsed_time(b) for _, a, b, _, _ in ev)
        return len(ev), ms, sum(e[3] for e in ev), sum(e[4] for e in ev)

    def launches(self, key=None) -> int:
        """every launch of the key(s) inside the profiled region, timed or not"""
        return self.seen[tuple(key)] if key is not None else sum(self.seen.values())


GEMM_PROFILE: Optional[GemmProfile] = None


def _out2d(M: int, N: int, like: torch.Tensor, out: Optional[torch.Tensor], out_dtype: Optional[torch.dtype]):
    if out is None:
        out = torch.empty((M, N), device=like.device, dtype=out_dtype or like.dtype)
    return out


def mm_nt(a: torch.Tensor, b: torch.Tensor, *, out=None, out_dtype=None, **kw) -> torch.Tensor:
    """out[M,N] = a[M,K] @ b[N,K]^T  (+ fused epilogue options of gemm())"""
    M, K = a.shape
    N = b.shape[0]
    assert b.shape[1] == K, (a.shape, b.shape)
    out = _out2d(M, N, a, out, out_dtype)
    kw = _ld_kwargs(kw, out)
    return gemm(L.NT, a, b, M, N, K, _row_major(a, "a"), _row_major(b, "b"), out, _row_major(out, "out"), **kw)


def mm_nn(a: torch.Tensor, b: torch.Tensor, *, out=None, out_dtype=None, **kw) -> torch.Tensor:
    """out[M,N] = a[M,K] @ b[K,N]"""
    M, K = a.shape
    N = b.shape[1]
    assert b.shape[0] == K, (a.shape, b.shape)
    out = _out2d(M, N, a, out, out_dtype)
    kw = _ld_kwargs(kw, out)
    return gemm(L.NN, a, b, M, N, K, _row_major(a, "a"), _row_major(b, "b"), out, _row_major(out, "out"), **kw)


def mm_tn(a: torch.Tensor, b: torch.Tensor, *, out=None, out_dtype=None, **kw) -> torch.Tensor:
    """out[M,N] = a[K,M]^T @ b[K,N]"""
    K, M = a.shape
    N = b.shape[1]
    assert b.shape[0] == K, (a.shape, b.shape)
    out = _out2d(M, N, a, out, out_dtype)
    kw = _ld_kwargs(kw, out)
    return gemm(L.TN, a, b, M, N, K, _row_major(a, "a"), _row_major(b, "b"), out, _row_major(out, "out"), **kw)


def swiglu_gemm_supported(x: torch.Tensor, w: torch.Tensor, keep_pre: bool = False) -> bool:
    """should / can mm_nt_swiglu take this gated-MLP product (dxa_gemm_desc.fuse = DXA_FUSE_SWIGLU)?  bf16, >= 129 rows, K % 64 == 0,
    F % 8 == 0, F >= 128.  Default: the serving paths (no pre-activations kept: one output stream); the training forward, which also
    stores the [M, 2F] pre-activations from the epilogue (three 8-byte store streams and 87 M SiLUs that nothing overlaps), measured
    235.5 against 236.1 ms per step with it — inside the noise, and the product's own rate drops — so it stays on the two-launch
    form unless DXA_SWIGLU_FUSE=1 (profiles/r06_swiglu_fuse_ab.txt).  DXA_SWIGLU_FUSE=0: two launches everywhere."""
    import os
    M, K_ = x.shape
    F_ = w.shape[0] // 2
    mode = os.environ.get("DXA_SWIGLU_FUSE", "")
    if mode == "0" or (keep_pre and mode != "1"):
        return False
    return (x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and x.is_cuda and
            M >= 129 and K_ % 64 == 0 and K_ >= 64 and F_ % 8 == 0 and F_ >= 128 and w.shape[0] % 2 == 0 and
            x.is_contiguous() and w.is_contiguous() and x.data_ptr() % 16 == 0 and w.data_ptr() % 16 == 0 and
            M * 2 * F_ * 2 < (1 << 31) and w.numel() * 2 < (1 << 31))


def mm_nt_swiglu(x: torch.Tensor, w: torch.Tensor, keep_pre: bool = True):
    """gated MLP input half in ONE launch: x [M, K] @ [gate ; up]^T ([2F, K], gate rows first) -> (silu(gate) * up [M, F],
    pre-activations [M, 2F] or None).  Bit-identical to ``swiglu_fwd(mm_nt(x, w))`` (same tiles, same K order, same rounding points);
    HF Qwen2MLP (qwen2/modeling_qwen2.py:35-48 under cogact_arch.py:97-106)."""
    M, K_ = x.shape
    N = w.shape[0]
    F_ = N // 2
    out = torch.empty((M, F_), device=x.device, dtype=x.dtype)
    pre = torch.empty((M, N), device=x.device, dtype=x.dtype) if keep_pre else None
    d = L.GemmDesc()
    d.layout, d.in_dtype, d.out_dtype, d.act = L.NT, L.BF16, L.BF16, L.ACT_NONE
    d.M, d.N, d.K = M, N, K_
    d.A, d.lda, d.B, d.ldb, d.C, d.ldc = _ptr(x), K_, _ptr(w), K_, _ptr(out), F_
    d.aux_out, d.ld_aux = _ptr(pre), N
    d.alpha, d.fuse = 1.0, L.FUSE_SWIGLU
    for i in range(3):
        d.nb[i] = 1
    prof = GEMM_PROFILE
    if prof is not None and prof.wants(L.NT, L.BF16, L.BF16):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.check(lib.dxa_gemm(C.byref(d), _stream()), "dxa_gemm (swiglu)")
        e1.record()
        prof.add((L.NT, L.BF16, L.BF16), e0, e1, 2.0 * M * N * K_, float((M * K_ + N * K_) * 2 + M * (N + F_) * 2))
        return out, pre
    L.check(lib.dxa_gemm(C.byref(d), _stream()), "dxa_gemm (swiglu)")
    return out, pre


def _ld_kwargs(kw: dict, out: torch.Tensor) -> dict:
    kw = dict(kw)
    if kw.get("residual") is not None:
        kw["ldr"] = _row_major(kw["residual"], "residual")
    if kw.get("mulgrad") is not None:
        kw["ldg"] = _row_major(kw["mulgrad"], "mulgrad")
    if kw.get("aux_out") is not None and _row_major(kw["aux_out"], "aux_out") != _row_major(out, "out"):
        raise L.DxaError("gemm: aux_out must share the output leading dimension")
    return kw


# ------------------------------------------------------------------------------------- MemVLA memory path
_MASK_STATE = {"seed": None, "offset": 0}


def dropout_mask(shape, p: float, dtype: torch.dtype, device, seed: Optional[int] = None) -> torch.Tensor:
    """a dropout mask (0 | 1/(1-p)) drawn on the device in ONE launch; successive masks advance a process-wide counter, the
    key comes from torch's seed (torch.manual_seed makes runs repeatable)"""
    n = 1
    for v in shape:
        n *= int(v)
    out = torch.empty(tuple(int(v) for v in shape), device=device, dtype=dtype)
    if _MASK_STATE["seed"] is None or seed is not None:
        _MASK_STATE["seed"] = int(torch.initial_seed() if seed is None else seed) & 0xFFFFFFFFFFFFFFFF
    _MASK_STATE["offset"] += 1
    L.check(lib.dxa_dropout_mask(_ptr(out), n, float(p), _MASK_STATE["seed"], _MASK_STATE["offset"], dt(out), _stream()),
            "dxa_dropout_mask")
    return out


def bank_consolidate(feat: torch.Tensor, ts: torch.Tensor, length: int, sims: torch.Tensor, fifo: bool = False) -> None:
    """token-merge of the most similar neighbouring pair (fifo: drop of the oldest entry) among the first ``length`` entries of
    feat [cap, N, D] / ts [cap] (in place; afterwards the first length - 1 entries are the bank)"""
    assert feat.is_contiguous() and ts.is_contiguous() and ts.dtype == torch.float32 and sims.dtype == torch.float32
    assert feat.dim() == 3 and length <= feat.shape[0] and sims.numel() >= length - 1
    L.check(lib.dxa_bank_consolidate(_ptr(feat), _ptr(ts), int(length), feat.shape[1], feat.shape[2], dt(feat), int(fifo),
                                     _ptr(sims), _stream()), "dxa_bank_consolidate")


def add_rows(x: Optional[torch.Tensor], g: torch.Tensor, Nn: int, alpha: float = 1.0) -> torch.Tensor:
    """x [R, Nn, C] + alpha * g [R, C] broadcast over the tokens (x None: the broadcast alone)"""
    R, C_ = g.shape
    assert g.is_contiguous() and (x is None or (x.is_contiguous() and x.shape == (R, Nn, C_) and x.dtype == g.dtype))
    out = torch.empty((R, Nn, C_), device=g.device, dtype=g.dtype)
    L.check(lib.dxa_add_rows(_ptr(x), _ptr(g), _ptr(out), R, Nn, C_, float(alpha), dt(g), _stream()), "dxa_add_rows")
    return out


def token_sum(x: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
    """scale * sum over the token axis: [R, Nn, C] -> [R, C] (same dtype, fp32 accumulation)"""
    R, Nn, C_ = x.shape
    assert x.is_contiguous()
    out = torch.empty((R, C_), device=x.device, dtype=x.dtype)
    L.check(lib.dxa_token_sum(_ptr(x), _ptr(out), R, Nn, C_, float(scale), dt(x), _stream()), "dxa_token_sum")
    return out


# ----------------------------------------------------------------------------------------------- norms
def rmsnorm_fwd(x: torch.Tensor, w: Optional[torch.Tensor], eps: float) -> Tuple[torch.Tensor, torch.Tensor]:
    x2 = x.reshape(-1, x.shape[-1])
    assert x2.is_contiguous()
    y = torch.empty_like(x2)
    rstd = torch.empty(x2.shape[0], device=x.device, dtype=torch.float32)
    L.check(lib.dxa_rmsnorm_fwd(_ptr(x2), _ptr(w), _ptr(y), _ptr(rstd), x2.shape[0], x2.shape[1], eps, dt(x2),
                                dt(w) if w is not None else dt(x2), _stream()), "dxa_rmsnorm_fwd")
    return y.view(x.shape), rstd


def norm_bwd_blocks(rows: int) -> int:
    return int(lib.dxa_norm_bwd_blocks(rows))


def _residual2d(residual, x2):
    if residual is None:
        return None
    r2 = residual.reshape(-1, x2.shape[-1])
    assert r2.is_contiguous() and r2.shape == x2.shape and r2.dtype == x2.dtype
    return r2


def rmsnorm_bwd(dy, x, w, rstd, dw_out: Optional[torch.Tensor] = None, accumulate: bool = False,
                want_dw: bool = True, residual: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """returns (dx, dw_fp32 or None); dw is (accumulated) into dw_out when given.  `residual` (the gradient that bypassed
    the normalised sub-block) is added to dx inside the kernel."""
    x2 = x.reshape(-1, x.shape[-1])
    dy2 = dy.reshape(-1, x.shape[-1])
    assert x2.is_contiguous() and dy2.is_contiguous()
    rows, cols = x2.shape
    dx = torch.empty_like(x2)
    part = None
    if w is not None and want_dw:
        part = torch.empty((norm_bwd_blocks(rows), cols), device=x.device, dtype=torch.float32)
    # w is still needed for dx even when its gradient is not (frozen norm): pass a throw-away slab
    if w is not None and part is None:
        part = torch.empty((norm_bwd_blocks(rows), cols), device=x.device, dtype=torch.float32)
        want_dw = False
    L.check(lib.dxa_rmsnorm_bwd(_ptr(dy2), _ptr(x2), _ptr(w), _ptr(rstd), _ptr(dx), _ptr(_residual2d(residual, x2)), _ptr(part),
                                rows, cols, dt(x2),
                                dt(w) if w is not None else dt(x2), _stream()), "dxa_rmsnorm_bwd")
    dw = colsum(part, out=dw_out, accumulate=accumulate) if (part is not None and want_dw) else None
    return dx.view(x.shape), dw


def layernorm_fwd(x, w, b, eps):
    x2 = x.reshape(-1, x.shape[-1])
    assert x2.is_contiguous()
    y = torch.empty_like(x2)
    mean = torch.empty(x2.shape[0], device=x.device, dtype=torch.float32)
    rstd = torch.empty_like(mean)
    L.check(lib.dxa_layernorm_fwd(_ptr(x2), _ptr(w), _ptr(b), _ptr(y), _ptr(mean), _ptr(rstd), x2.shape[0],
                                  x2.shape[1], eps, dt(x2), dt(w) if w is not None else dt(x2), _stream()),
            "dxa_layernorm_fwd")
    return y.view(x.shape), mean, rstd


def layernorm_bwd(dy, x, w, mean, rstd, dw_out=None, db_out=None, accumulate: bool = False, want_dw: bool = True,
                  residual: Optional[torch.Tensor] = None, return_part: bool = False):
    """returns (dx, dw_fp32 or None, db_fp32 or None); dw/db are (accumulated) into dw_out/db_out when given; `residual` is
    added to dx inside the kernel.  ``return_part``: (dx, the kernel's per-row-block partial sums [blocks, dw | db]) — the caller folds them"""
    x2 = x.reshape(-1, x.shape[-1])
    dy2 = dy.reshape(-1, x.shape[-1])
    assert x2.is_contiguous() and dy2.is_contiguous()
    rows, cols = x2.shape
    dx = torch.empty_like(x2)
    part = None
    if w is not None:
        part = torch.empty((norm_bwd_blocks(rows), 2 * cols), device=x.device, dtype=torch.float32)
    L.check(lib.dxa_layernorm_bwd(_ptr(dy2), _ptr(x2), _ptr(w), _ptr(mean), _ptr(rstd), _ptr(dx),
                                  _ptr(_residual2d(residual, x2)), _ptr(part), rows, cols, dt(x2), dt(w) if w is not None else dt(x2), _stream()), "dxa_layernorm_bwd")
    if return_part:
        return dx.view(x.shape), part
    if part is None or not want_dw:
        return dx.view(x.shape), None, None
    if dw_out is not None:
        if db_out is not None and db_out.data_ptr() == dw_out.data_ptr() + 4 * cols:
            # weight and bias gradient slots lie back to back (they are registered as one group): one column sum
            both = torch.as_strided(dw_out, (2 * cols,), (1,))
            colsum(part, out=both, accumulate=accumulate)
        else:
            colsum(part[:, :cols], out=dw_out, accumulate=accumulate)
            colsum(part[:, cols:], out=db_out, accumulate=accumulate)
        return dx.view(x.shape), dw_out, db_out
    s = colsum(part)
    return dx.view(x.shape), s[:cols], s[cols:]


def colsum(x: torch.Tensor, out: Optional[torch.Tensor] = None, accumulate: bool = False) -> torch.Tensor:
    """out[c] (+)= sum_r x[r, c]  (fp32 result)"""
    ld = _row_major(x, "x")
    rows, cols = x.shape
    if out is None:
        out = torch.empty(cols, device=x.device, dtype=torch.float32)
        accumulate = False
    nsplit = max(1, min(64, (rows + 31) // 32))
    scratch = torch.empty(nsplit * cols, device=x.device, dtype=torch.float32)
    L.check(lib.dxa_colsum(_ptr(x), ld, _ptr(out), rows, cols, dt(x), int(accumulate), _ptr(scratch),
                           scratch.numel() * 4, _stream()), "dxa_colsum")
    return out


# ------------------------------------------------------------------------------------------------ RoPE
def rope_split(qkv: torch.Tensor, cos_t, sin_t, pos, B, S, Hq, Hkv, D, k_out=None, v_out=None):
    """k_out / v_out: contiguous [B, Hkv, S, D] destinations (e.g. the tail of a key/value cache) instead of fresh tensors"""
    assert qkv.is_contiguous() and qkv.shape == (B * S, (Hq + 2 * Hkv) * D)
    q = torch.empty((B, Hq, S, D), device=qkv.device, dtype=qkv.dtype)
    k = torch.empty((B, Hkv, S, D), device=qkv.device, dtype=qkv.dtype) if k_out is None else k_out
    v = torch.empty((B, Hkv, S, D), device=qkv.device, dtype=qkv.dtype) if v_out is None else v_out
    assert k.is_contiguous() and v.is_contiguous() and k.shape == (B, Hkv, S, D) == v.shape and k.dtype == qkv.dtype == v.dtype
    L.check(lib.dxa_rope_split(_ptr(qkv), _ptr(q), _ptr(k), _ptr(v), _ptr(cos_t), _ptr(sin_t), _ptr(pos), B, S, Hq,
                               Hkv, D, dt(qkv), _stream()), "dxa_rope_split")
    return q, k, v


def rope_split_into(qkv: torch.Tensor, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, s0: int, cos_t, sin_t, pos, B, S, Hq, Hkv, D) -> None:
    """rope_split writing its S tokens at positions s0 .. s0 + S - 1 of SHARED contiguous q [B, Hq, S_cap, D], k / v [B, Hkv, S_cap, D]"""
    S_cap = q.shape[2]
    assert qkv.is_contiguous() and qkv.shape == (B * S, (Hq + 2 * Hkv) * D)
    assert q.is_contiguous() and k.is_contiguous() and v.is_contiguous() and q.shape == (B, Hq, S_cap, D) and \
        k.shape == (B, Hkv, S_cap, D) == v.shape and q.dtype == k.dtype == v.dtype == qkv.dtype
    L.check(lib.dxa_rope_split_at(_ptr(qkv), _ptr(q), _ptr(k), _ptr(v), _ptr(cos_t), _ptr(sin_t), _ptr(pos), B, S, Hq, Hkv, D,
                                  S_cap, s0, dt(qkv), _stream()), "dxa_rope_split_at")


def rope_merge_from(dq, dk, dv, s0: int, cos_t, sin_t, pos, B, S, Hq, Hkv, D):
    """rope_merge of positions s0 .. s0 + S - 1 of SHARED contiguous dq [B, Hq, S_cap, D], dk / dv [B, Hkv, S_cap, D]"""
    S_cap = dq.shape[2]
    assert dq.is_contiguous() and dk.is_contiguous() and dv.is_contiguous() and dq.shape == (B, Hq, S_cap, D)
    dqkv = torch.empty((B * S, (Hq + 2 * Hkv) * D), device=dq.device, dtype=dq.dtype)
    L.check(lib.dxa_rope_merge_at(_ptr(dq), _ptr(dk), _ptr(dv), _ptr(dqkv), _ptr(cos_t), _ptr(sin_t), _ptr(pos), B, S, Hq, Hkv, D,
                                  S_cap, s0, dt(dq), _stream()), "dxa_rope_merge_at")
    return dqkv


def rope_merge(dq, dk, dv, cos_t, sin_t, pos, B, S, Hq, Hkv, D):
    assert dq.is_contiguous() and dk.is_contiguous() and dv.is_contiguous()
    dqkv = torch.empty((B * S, (Hq + 2 * Hkv) * D), device=dq.device, dtype=dq.dtype)
    L.check(lib.dxa_rope_merge(_ptr(dq), _ptr(dk), _ptr(dv), _ptr(dqkv), _ptr(cos_t), _ptr(sin_t), _ptr(pos), B, S,
                               Hq, Hkv, D, dt(dq), _stream()), "dxa_rope_merge")
    return dqkv


# ------------------------------------------------------------------------------------------- attention
def _bhsd_strides(t: torch.Tensor) -> Tuple[int, int, int]:
    """t is a [B,H,S,D] VIEW (any memory order) with D contiguous -> (sb, sh, ss)"""
    assert t.dim() == 4 and (t.shape[3] == 1 or t.stride(3) == 1), (t.shape, t.stride())
    return t.stride(0), t.stride(1), t.stride(2)


def _attn_desc(q, k, v, o, lse, causal, scale, kv_start, kv_end, q_limit=None, key_valid=None, drop_mask=None) -> L.AttnDesc:
    d = L.AttnDesc()
    B, Hq, Sq, D = q.shape
    _, Hkv, Sk, _ = k.shape
    d.dtype, d.B, d.Hq, d.Hkv, d.Sq, d.Sk, d.D = dt(q), B, Hq, Hkv, Sq, Sk, D
    d.causal, d.scale = int(causal), scale
    d.q, (d.q_sb, d.q_sh, d.q_ss) = _ptr(q), _bhsd_strides(q)
    d.k, (d.k_sb, d.k_sh, d.k_ss) = _ptr(k), _bhsd_strides(k)
    d.v, (d.v_sb, d.v_sh, d.v_ss) = _ptr(v), _bhsd_strides(v)
    d.o, (d.o_sb, d.o_sh, d.o_ss) = _ptr(o), _bhsd_strides(o)
    d.lse, d.kv_start, d.kv_end = _ptr(lse), _ptr(kv_start), _ptr(kv_end)
    if q_limit is not None:
        assert q_limit.dtype == torch.int32 and q_limit.is_contiguous() and q_limit.shape == (B, Sq)
    if key_valid is not None:
        assert key_valid.dtype == torch.uint8 and key_valid.is_contiguous() and key_valid.shape == (B, Sk)
    d.q_limit, d.key_valid = _ptr(q_limit), _ptr(key_valid)
    if drop_mask is not None:
        assert drop_mask.dtype == q.dtype and drop_mask.is_contiguous() and drop_mask.shape == (B, Hq, Sq, Sk)
        d.drop_mask = _ptr(drop_mask)
    return d


def attn_fwd(q, k, v, o, *, causal: bool, scale: float, kv_start=None, kv_end=None, force_generic=False,
             q_limit=None, key_valid=None, drop_mask=None):
    """q/k/v/o: [B,H,S,D] views (token-major or head-major memory); returns lse [B,Hq,Sq] fp32"""
    B, Hq, Sq, _ = q.shape
    lse = torch.empty((B, Hq, Sq), device=q.device, dtype=torch.float32)
    d = _attn_desc(q, k, v, o, lse, causal, scale, kv_start, kv_end, q_limit, key_valid, drop_mask)
    d.force_generic = int(force_generic)
    nbytes = int(lib.dxa_attn_fwd_workspace(C.byref(d)))
    if nbytes:
        # large non-flash problem (fp32, or attention dropout): scores materialised through the batched GEMMs
        ws = torch.empty(nbytes, device=q.device, dtype=torch.uint8)
        L.check(lib.dxa_attn_fwd_ws(C.byref(d), _ptr(ws), nbytes, _stream()), "dxa_attn_fwd_ws")
    else:
        L.check(lib.dxa_attn_fwd(C.byref(d), _stream()), "dxa_attn_fwd")
    return lse


def attn_bwd(q, k, v, o, lse, do, dq, dk, dv, *, causal: bool, scale: float, kv_start=None, kv_end=None,
             force_generic=False, q_limit=None, key_valid=None, drop_mask=None):
    d = _attn_desc(q, k, v, o, lse, causal, scale, kv_start, kv_end, q_limit, key_valid, drop_mask)
    d.force_generic = int(force_generic)
    d.d_o, (d.do_sb, d.do_sh, d.do_ss) = _ptr(do), _bhsd_strides(do)
    d.dq, (d.dq_sb, d.dq_sh, d.dq_ss) = _ptr(dq), _bhsd_strides(dq)
    d.dk, (d.dk_sb, d.dk_sh, d.dk_ss) = _ptr(dk), _bhsd_strides(dk)
    d.dv, (d.dv_sb, d.dv_sh, d.dv_ss) = _ptr(dv), _bhsd_strides(dv)
    nbytes = int(lib.dxa_attn_bwd_workspace(C.byref(d)))
    ws = torch.empty(nbytes, device=q.device, dtype=torch.uint8)
    L.check(lib.dxa_attn_bwd(C.byref(d), _ptr(ws), nbytes, _stream()), "dxa_attn_bwd")


# ----------------------------------------------------------------------------------------- elementwise
def swiglu_fwd(gu: torch.Tensor) -> torch.Tensor:
    rows, F2 = gu.shape
    assert gu.is_contiguous() and F2 % 2 == 0
    out = torch.empty((rows, F2 // 2), device=gu.device, dtype=gu.dtype)
    L.check(lib.dxa_swiglu_fwd(_ptr(gu), _ptr(out), rows, F2 // 2, dt(gu), _stream()), "dxa_swiglu_fwd")
    return out


def swiglu_bwd(gu: torch.Tensor, dout: torch.Tensor) -> torch.Tensor:
    rows, F2 = gu.shape
    assert gu.is_contiguous() and dout.is_contiguous()
    dgu = torch.empty_like(gu)
    L.check(lib.dxa_swiglu_bwd(_ptr(gu), _ptr(dout), _ptr(dgu), rows, F2 // 2, dt(gu), _stream()), "dxa_swiglu_bwd")
    return dgu


def axpby(a: torch.Tensor, b: torch.Tensor, alpha: float, beta: float) -> torch.Tensor:
    assert a.is_contiguous() and b.is_contiguous() and a.shape == b.shape and a.dtype == b.dtype
    out = torch.empty_like(a)
    L.check(lib.dxa_axpby(_ptr(a), _ptr(b), _ptr(out), a.numel(), alpha, beta, dt(a), _stream()), "dxa_axpby")
    return out


def mul(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    assert a.is_contiguous() and b.is_contiguous() and a.shape == b.shape and a.dtype == b.dtype
    out = torch.empty_like(a)
    L.check(lib.dxa_mul(_ptr(a), _ptr(b), _ptr(out), a.numel(), dt(a), _stream()), "dxa_mul")
    return out


def mul_rows(x: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    """x [R, N, C] * g [R, C] (broadcast over N)"""
    R, Nn, C_ = x.shape
    assert x.is_contiguous() and g.is_contiguous() and g.shape == (R, C_) and g.dtype == x.dtype
    out = torch.empty_like(x)
    L.check(lib.dxa_mul_rows(_ptr(x), _ptr(g), _ptr(out), R, Nn, C_, dt(x), _stream()), "dxa_mul_rows")
    return out


def glu_fwd(gu: torch.Tensor, act: int) -> torch.Tensor:
    """act(gate) * up for [rows, 2F] = [gate ; up] (GeGLU with act = ACT_GELU_TANH)"""
    rows, F2 = gu.shape
    assert gu.is_contiguous() and F2 % 2 == 0
    out = torch.empty((rows, F2 // 2), device=gu.device, dtype=gu.dtype)
    L.check(lib.dxa_glu_fwd(_ptr(gu), _ptr(out), rows, F2 // 2, act, dt(gu), _stream()), "dxa_glu_fwd")
    return out


def glu_bwd(gu: torch.Tensor, dout: torch.Tensor, act: int) -> torch.Tensor:
    rows, F2 = gu.shape
    assert gu.is_contiguous() and dout.is_contiguous()
    dgu = torch.empty_like(gu)
    L.check(lib.dxa_glu_bwd(_ptr(gu), _ptr(dout), _ptr(dgu), rows, F2 // 2, act, dt(gu), _stream()), "dxa_glu_bwd")
    return dgu


def act_fwd(x: torch.Tensor, act: int) -> torch.Tensor:
    assert x.is_contiguous()
    y = torch.empty_like(x)
    L.check(lib.dxa_act_fwd(_ptr(x), _ptr(y), x.numel(), act, dt(x), _stream()), "dxa_act_fwd")
    return y


def act_bwd(x: torch.Tensor, dy: torch.Tensor, act: int) -> torch.Tensor:
    assert x.is_contiguous() and dy.is_contiguous()
    dx = torch.empty_like(x)
    L.check(lib.dxa_act_bwd(_ptr(x), _ptr(dy), _ptr(dx), x.numel(), act, dt(x), _stream()), "dxa_act_bwd")
    return dx


def add(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    assert a.is_contiguous() and b.is_contiguous() and a.shape == b.shape and a.dtype == b.dtype
    if out is None:
        out = torch.empty_like(a)
    L.check(lib.dxa_add(_ptr(a), _ptr(b), _ptr(out), a.numel(), dt(a), _stream()), "dxa_add")
    return out


def cast(src: torch.Tensor, dtype: torch.dtype, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    assert src.is_contiguous()
    if out is None:
        out = torch.empty(src.shape, device=src.device, dtype=dtype)
    L.check(lib.dxa_cast(_ptr(src), _ptr(out), src.numel(), dt(src), dt(out), _stream()), "dxa_cast")
    return out


def copy2d(src: torch.Tensor, dst: torch.Tensor, cols: int, cols_padded: int) -> torch.Tensor:
    L.check(lib.dxa_copy2d(_ptr(src), _row_major(src, "src"), _ptr(dst), _row_major(dst, "dst"), src.shape[0], cols,
                           cols_padded, dt(src), dt(dst), _stream()), "dxa_copy2d")
    return dst


def transpose(x: torch.Tensor, pad_to: int = 1, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[R, C] -> [C, Rp] with Rp = R rounded up to `pad_to` (zero filled)"""
    R, C_ = x.shape
    Rp = (R + pad_to - 1) // pad_to * pad_to
    if out is None:
        out = torch.empty((C_, Rp), device=x.device, dtype=x.dtype)
    L.check(lib.dxa_transpose(_ptr(x), _row_major(x, "x"), _ptr(out), _row_major(out, "out"), R, C_, Rp, dt(x), _stream()),
            "dxa_transpose")
    return out


def permute_bshd(x: torch.Tensor, B: int, S: int, H: int, D: int, to_head: bool) -> torch.Tensor:
    assert x.is_contiguous() and x.numel() == B * S * H * D
    out = torch.empty((B, H, S, D) if to_head else (B, S, H, D), device=x.device, dtype=x.dtype)
    L.check(lib.dxa_permute_bshd(_ptr(x), _ptr(out), B, S, H, D, int(to_head), dt(x), _stream()), "dxa_permute_bshd")
    return out


def splice_fwd(plan: torch.Tensor, embed: torch.Tensor, img: Optional[torch.Tensor]) -> torch.Tensor:
    assert plan.dtype == torch.int64 and plan.is_contiguous() and embed.is_contiguous()
    n, d = plan.numel(), embed.shape[1]
    out = torch.empty((n, d), device=embed.device, dtype=embed.dtype)
    L.check(lib.dxa_splice_fwd(_ptr(plan), _ptr(embed), _ptr(img), _ptr(out), n, d, dt(embed), _stream()),
            "dxa_splice_fwd")
    return out


def splice_bwd(plan: torch.Tensor, dout: torch.Tensor, d_embed: Optional[torch.Tensor],
               d_img: Optional[torch.Tensor]) -> None:
    n, d = dout.shape
    assert dout.is_contiguous() and (d_embed is None or d_embed.dtype == torch.float32)
    L.check(lib.dxa_splice_bwd(_ptr(plan), _ptr(dout), _ptr(d_embed), _ptr(d_img), n, d, dt(dout), _stream()),
            "dxa_splice_bwd")


def zero_rows(plan: torch.Tensor, g: torch.Tensor) -> None:
    """g[plan[r], :] = 0 for every token entry (plan[r] >= 0) of the plan; g fp32 [V, d]"""
    assert plan.dtype == torch.int64 and plan.is_contiguous() and g.dtype == torch.float32 and g.is_contiguous()
    L.check(lib.dxa_zero_rows(_ptr(plan), _ptr(g), plan.numel(), g.shape[1], _stream()), "dxa_zero_rows")


def gather_rows(x: torch.Tensor, idx: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    assert x.is_contiguous() and idx.dtype == torch.int64
    out = torch.empty((idx.numel(), x.shape[1]), device=x.device, dtype=dtype)
    L.check(lib.dxa_gather_rows(_ptr(x), _ptr(idx), _ptr(out), idx.numel(), x.shape[1], dt(x), dt(out), _stream()),
            "dxa_gather_rows")
    return out


def scatter_rows(dout: torch.Tensor, idx: torch.Tensor, R: int, dtype: torch.dtype) -> torch.Tensor:
    assert dout.is_contiguous()
    dx = torch.empty((R, dout.shape[1]), device=dout.device, dtype=dtype)
    L.check(lib.dxa_scatter_rows(_ptr(dout), _ptr(idx), _ptr(dx), idx.numel(), R, dout.shape[1], dt(dout), dt(dx),
                                 _stream()), "dxa_scatter_rows")
    return dx


def im2col(images: torch.Tensor, P: int, ld: int, dtype: torch.dtype) -> torch.Tensor:
    assert images.is_contiguous() and images.dim() == 4 and images.shape[1] == 3
    N, _, H, W = images.shape
    rows = torch.empty((N * (H // P) * (W // P), ld), device=images.device, dtype=dtype)
    L.check(lib.dxa_im2col(_ptr(images), _ptr(rows), N, H, W, P, ld, dt(images), dt(rows), _stream()), "dxa_im2col")
    return rows


def vit_embed_fwd(patch: torch.Tensor, cls: torch.Tensor, pos: torch.Tensor, N: int, np_: int) -> torch.Tensor:
    C_ = patch.shape[-1]
    x = torch.empty((N, np_ + 1, C_), device=patch.device, dtype=patch.dtype)
    L.check(lib.dxa_vit_embed_fwd(_ptr(patch), _ptr(cls), _ptr(pos), _ptr(x), N, np_, C_, dt(patch), dt(cls),
                                  _stream()), "dxa_vit_embed_fwd")
    return x


def vit_embed_bwd(dx: torch.Tensor, N: int, np_: int) -> torch.Tensor:
    C_ = dx.shape[-1]
    assert dx.is_contiguous()
    dpatch = torch.empty((N * np_, C_), device=dx.device, dtype=dx.dtype)
    L.check(lib.dxa_vit_embed_bwd(_ptr(dx), _ptr(dpatch), N, np_, C_, dt(dx), _stream()), "dxa_vit_embed_bwd")
    return dpatch


# ------------------------------------------------------------------------------------- diffusion glue
def qsample(x0, noise, a, s):
    assert all(t.dtype == torch.float32 and t.is_contiguous() for t in (x0, noise, a, s))
    xt = torch.empty_like(x0)
    N = x0.shape[0]
    L.check(lib.dxa_qsample(_ptr(x0), _ptr(noise), _ptr(a), _ptr(s), _ptr(xt), N, x0.numel() // N, _stream()),
            "dxa_qsample")
    return xt


def timestep_embedding(t: torch.Tensor, freqs: torch.Tensor) -> torch.Tensor:
    assert t.dtype == torch.float32 and freqs.dtype == torch.float32
    half = freqs.numel()
    out = torch.empty((t.numel(), 2 * half), device=t.device, dtype=torch.float32)
    L.check(lib.dxa_timestep_embedding(_ptr(t), _ptr(freqs), _ptr(out), t.numel(), half, _stream()),
            "dxa_timestep_embedding")
    return out


def dit_assemble_fwd(xe, te, ze, pos):
    N, T, hd = xe.shape
    h = torch.empty((N, T + 1, hd), device=xe.device, dtype=torch.float32)
    L.check(lib.dxa_dit_assemble_fwd(_ptr(xe), _ptr(te), _ptr(ze), _ptr(pos), _ptr(h), N, T, hd, _stream()),
            "dxa_dit_assemble_fwd")
    return h


def dit_assemble_bwd(dh):
    N, T1, hd = dh.shape
    assert dh.is_contiguous()
    dxe = torch.empty((N, T1 - 1, hd), device=dh.device, dtype=torch.float32)
    dc = torch.empty((N, hd), device=dh.device, dtype=torch.float32)
    L.check(lib.dxa_dit_assemble_bwd(_ptr(dh), _ptr(dxe), _ptr(dc), N, T1 - 1, hd, _stream()), "dxa_dit_assemble_bwd")
    return dxe, dc


def token_drop(z, uncond, drop):
    N, d = z.shape
    assert drop.dtype == torch.uint8 and z.is_contiguous()
    out = torch.empty_like(z)
    L.check(lib.dxa_token_drop(_ptr(z), _ptr(uncond), _ptr(drop), _ptr(out), N, d, _stream()), "dxa_token_drop")
    return out


def token_drop_bwd(dout, drop, want_dz=True):
    N, d = dout.shape
    assert dout.is_contiguous()
    dz = torch.empty_like(dout) if want_dz else None
    dunc = torch.empty(d, device=dout.device, dtype=torch.float32)
    L.check(lib.dxa_token_drop_bwd(_ptr(dout), _ptr(drop), _ptr(dz), _ptr(dunc), N, d, 0, _stream()),
            "dxa_token_drop_bwd")
    return dz, dunc


def mse_loss(pred, target, gscale: float = 1.0, want_grad: bool = True):
    assert pred.is_contiguous() and target.is_contiguous() and pred.dtype == torch.float32
    loss = torch.empty(1, device=pred.device, dtype=torch.float32)
    dpred = torch.empty_like(pred) if want_grad else None
    L.check(lib.dxa_mse_loss(_ptr(pred), _ptr(target), _ptr(loss), _ptr(dpred), pred.numel(), gscale, _stream()),
            "dxa_mse_loss")
    return loss, dpred


def mse_loss_rows(pred, target, row_w, gscale: float = 1.0, want_grad: bool = True):
    """sum_r w_r mean_c(d^2) / (sum w + 1e-6) over pred/target [rows, ...]; row_w [rows] fp32"""
    assert pred.is_contiguous() and target.is_contiguous() and pred.dtype == torch.float32
    rows = pred.shape[0]
    assert row_w.dtype == torch.float32 and row_w.is_contiguous() and row_w.numel() == rows
    loss = torch.empty(1, device=pred.device, dtype=torch.float32)
    dpred = torch.empty_like(pred) if want_grad else None
    L.check(lib.dxa_mse_loss_rows(_ptr(pred), _ptr(target), _ptr(row_w), _ptr(loss), _ptr(dpred), rows,
                                  pred.numel() // rows, gscale, _stream()), "dxa_mse_loss_rows")
    return loss, dpred


def ddim_step(x, model_out, B, use_cfg, cfg_scale, c_recip, c_recipm1, ab_prev):
    per = x.numel() // x.shape[0]
    L.check(lib.dxa_ddim_step(_ptr(x), _ptr(model_out), B, per, int(use_cfg), cfg_scale, c_recip, c_recipm1, ab_prev,
                              _stream()), "dxa_ddim_step")
    return x


# --------------------------------------------------------------------------------------------- optimizer
def sumsq(x: torch.Tensor, out: torch.Tensor, scratch: torch.Tensor, accumulate: bool = False) -> torch.Tensor:
    assert x.is_contiguous() and scratch.dtype == torch.float64 and scratch.numel() >= 4096
    L.check(lib.dxa_sumsq(_ptr(x), x.numel(), dt(x), _ptr(scratch), _ptr(out), int(accumulate), _stream()), "dxa_sumsq")
    return out


def sumsq_ranges(base: torch.Tensor, starts: torch.Tensor, lens: torch.Tensor, out: torch.Tensor, scratch: torch.Tensor,
                 accumulate: bool = False) -> torch.Tensor:
    """out (+)= sum over the slices base[starts[i] : starts[i] + lens[i]] of x^2 (int64 device arrays, <= 4096 slices)"""
    assert base.is_contiguous() and starts.dtype == torch.int64 and lens.dtype == torch.int64 and starts.numel() == lens.numel()
    assert scratch.dtype == torch.float64 and scratch.numel() >= 4096
    L.check(lib.dxa_sumsq_ranges(_ptr(base), dt(base), _ptr(starts), _ptr(lens), starts.numel(), _ptr(scratch), _ptr(out),
                                 int(accumulate), _stream()), "dxa_sumsq_ranges")
    return out


def sum_f32(x: torch.Tensor, out: torch.Tensor, accumulate: bool = False) -> torch.Tensor:
    assert x.is_contiguous() and x.dtype == torch.float32 and out.dtype == torch.float32
    L.check(lib.dxa_sum_f32(_ptr(x), x.numel(), _ptr(out), int(accumulate), _stream()), "dxa_sum_f32")
    return out


def gemm_sumsq_slots(M: int, N: int) -> int:
    return int(lib.dxa_gemm_sumsq_slots(M, N))


def clip_coef(sumsq_t, max_norm: float, norm_out, coef_out, grad_scale: float = 1.0):
    if grad_scale == 1.0:
        L.check(lib.dxa_clip_coef(_ptr(sumsq_t), max_norm, _ptr(norm_out), _ptr(coef_out), _stream()), "dxa_clip_coef")
    else:
        L.check(lib.dxa_clip_coef_scaled(_ptr(sumsq_t), max_norm, grad_scale, _ptr(norm_out), _ptr(coef_out), _stream()),
                "dxa_clip_coef_scaled")


def scale_(x: torch.Tensor, s: float) -> torch.Tensor:
    assert x.dtype == torch.float32 and x.is_contiguous()
    L.check(lib.dxa_scale(_ptr(x), x.numel(), s, _stream()), "dxa_scale")
    return x


def scale_dev_(x: torch.Tensor, s: torch.Tensor) -> torch.Tensor:
    """x *= s[0], s a device fp32 scalar"""
    assert x.dtype == torch.float32 and x.is_contiguous() and s.dtype == torch.float32 and s.numel() == 1
    L.check(lib.dxa_scale_dev(_ptr(x), x.numel(), _ptr(s), _stream()), "dxa_scale_dev")
    return x


def adamw(p, g, m, v, shadow, chunk_start, chunk_len, chunk_grp, lrs, wds, beta1, beta2, eps, step, clip=None, chunk_state=None,
          chunk_mv_start=None):
    """``chunk_mv_start``: m / v are PACKED (the sharded optimizer state of engine.FusedAdamW(ranges=...)): chunk c's moments start
    at element chunk_mv_start[c] of m and v; p, g and shadow keep the arena offset chunk_start[c]"""
    d = L.AdamWDesc()
    if chunk_mv_start is not None:
        assert chunk_mv_start.dtype == torch.int64 and chunk_mv_start.numel() == chunk_start.numel() and chunk_mv_start.is_contiguous()
    d.chunk_mv_start = _ptr(chunk_mv_start)
    if chunk_state is not None:
        assert chunk_state.dtype == torch.uint8 and chunk_state.numel() == chunk_start.numel() and chunk_state.is_contiguous()
    d.chunk_state = _ptr(chunk_state)
    d.p, d.g, d.m, d.v, d.shadow = _ptr(p), _ptr(g), _ptr(m), _ptr(v), _ptr(shadow)
    d.g_dtype = dt(g)
    d.chunk_start, d.chunk_len, d.chunk_grp = _ptr(chunk_start), _ptr(chunk_len), _ptr(chunk_grp)
    d.n_chunks = chunk_start.numel()
    assert len(lrs) <= 8 and len(lrs) == len(wds)
    for i, (lr, wd) in enumerate(zip(lrs, wds)):
        d.lr[i], d.wd[i] = lr, wd
    d.beta1, d.beta2, d.eps = beta1, beta2, eps
    d.bc1, d.bc2 = 1.0 - beta1 ** step, 1.0 - beta2 ** step
    d.clip_coef = _ptr(clip)
    L.check(lib.dxa_adamw(C.byref(d), _stream()), "dxa_adamw")


# ------------------------------------------------------------------------------------- LM head (row A10)
def cross_entropy_fwd(logits: torch.Tensor, labels: torch.Tensor, ignore_index: int = -100):
    """logits [rows, V] (fp32/bf16, rows contiguous), labels [rows] int64 (already shifted) ->
    (row_loss [rows] fp32, lse [rows] fp32)"""
    rows, V = logits.shape
    assert logits.stride(1) == 1 and labels.dtype == torch.int64 and labels.is_contiguous() and labels.numel() == rows
    row_loss = torch.empty(rows, device=logits.device, dtype=torch.float32)
    lse = torch.empty(rows, device=logits.device, dtype=torch.float32)
    L.check(lib.dxa_cross_entropy_fwd(_ptr(logits), logits.stride(0), _ptr(labels), _ptr(row_loss), _ptr(lse), rows, V,
                                      ignore_index, dt(logits), _stream()), "dxa_cross_entropy_fwd")
    return row_loss, lse


def cross_entropy_bwd(logits: torch.Tensor, labels: torch.Tensor, lse: torch.Tensor, gscale: Optional[torch.Tensor],
                      scale: float, out: Optional[torch.Tensor] = None, ignore_index: int = -100) -> torch.Tensor:
    """dlogits = (softmax - onehot) * gscale[0] * scale; ``out`` may be ``logits`` itself (in place)"""
    rows, V = logits.shape
    if out is None:
        out = torch.empty_like(logits)
    assert out.shape == logits.shape and out.dtype == logits.dtype and out.stride(1) == 1
    L.check(lib.dxa_cross_entropy_bwd(_ptr(logits), logits.stride(0), _ptr(labels), _ptr(lse), _ptr(gscale), float(scale),
                                      _ptr(out), out.stride(0), rows, V, ignore_index, dt(logits), _stream()),
            "dxa_cross_entropy_bwd")
    return out


def argmax_rows(x: torch.Tensor) -> torch.Tensor:
    """first index of each row's maximum (torch.argmax semantics) -> int64 [rows]"""
    rows, cols = x.shape
    assert x.stride(1) == 1
    out = torch.empty(rows, device=x.device, dtype=torch.int64)
    L.check(lib.dxa_argmax_rows(_ptr(x), x.stride(0), _ptr(out), rows, cols, dt(x), _stream()), "dxa_argmax_rows")
    return out


# ------------------------------------------------------------------------------------------------ image preprocessing
_RESAMPLE_TABLES: dict = {}


def resample_tables(in_size: int, out_size: int, device) -> Tuple[int, "torch.Tensor", "torch.Tensor", "torch.Tensor"]:
    """Pillow bicubic tap tables of one axis: (ksize, bounds host int32 [out,2], bounds device, taps device [out,ksize]);
    built once per (in_size, out_size, device) by the library's host routine and kept on the device"""
    key = (int(in_size), int(out_size), str(device))
    hit = _RESAMPLE_TABLES.get(key)
    if hit is None:
        ks = lib.dxa_resample_ksize(int(in_size), int(out_size))
        bounds = torch.empty(out_size, 2, dtype=torch.int32)
        taps = torch.empty(out_size, ks, dtype=torch.int32)
        L.check(lib.dxa_resample_coeffs(int(in_size), int(out_size), L.FILTER_BICUBIC, bounds.data_ptr(), taps.data_ptr()),
                "dxa_resample_coeffs")
        hit = (ks, bounds, bounds.to(device), taps.to(device))
        _RESAMPLE_TABLES[key] = hit
    return hit


def resize_output_size(h: int, w: int, shortest_edge: int) -> Tuple[int, int]:
    """(out_h, out_w) of the processor's shortest-edge resize: long side = int(shortest_edge * long / short)"""
    if w <= h:
        return int(shortest_edge * h / w), shortest_edge
    return shortest_edge, int(shortest_edge * w / h)


def image_preprocess(frames: torch.Tensor, *, pad: bool, bg: Sequence[int], size: int, crop: Tuple[int, int],
                     mean: Sequence[float], std: Sequence[float], rescale: float = 1 / 255,
                     out_dtype: torch.dtype = torch.float32, want_u8: bool = False):
    """uint8 RGB frames [n, h, w, 3] on the device -> [n, 3, crop_h, crop_w] (expand2square, Pillow-exact bicubic
    shortest-edge resize, center crop, rescale, normalise).  With want_u8 also returns the uint8 resized frame."""
    if frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[-1] != 3 or not frames.is_contiguous():
        raise L.DxaError(f"image_preprocess: expected contiguous uint8 [n,h,w,3], got {frames.dtype} {tuple(frames.shape)}")
    n, h, w, _ = frames.shape
    ph, pw = (max(h, w), max(h, w)) if pad else (h, w)
    res_h, res_w = resize_output_size(ph, pw, size)
    ch, cw = crop
    if ch > res_h or cw > res_w:
        raise L.DxaError(f"image_preprocess: crop {crop} larger than the resized frame {(res_h, res_w)}")
    top, left = (res_h - ch) // 2, (res_w - cw) // 2
    dev = frames.device
    d = L.ImageDesc()
    d.src, d.n, d.h, d.w, d.pad = _ptr(frames), n, h, w, int(bool(pad))
    for i in range(3):
        d.bg[i] = int(bg[i])
        d.mean[i] = float(mean[i])
        d.std[i] = float(std[i])
    d.res_h, d.res_w, d.crop_top, d.crop_left, d.out_h, d.out_w = res_h, res_w, top, left, ch, cw
    keep = []
    if res_w != pw:
        d.hks, _, hb, hk = resample_tables(pw, res_w, dev)
        d.hb, d.hk = _ptr(hb), _ptr(hk)
        keep += [hb, hk]
    row0, rows = top, ch
    if res_h != ph:
        d.vks, vb_host, vb, vk = resample_tables(ph, res_h, dev)
        d.vb, d.vk = _ptr(vb), _ptr(vk)
        keep += [vb, vk]
        win = vb_host[top:top + ch]
        row0 = int(win[:, 0].min())
        rows = int((win[:, 0] + win[:, 1]).max()) - row0
    d.row0, d.rows = row0, rows
    tmp = torch.empty(n * rows * cw * 3, device=dev, dtype=torch.uint8)
    out = torch.empty(n, 3, ch, cw, device=dev, dtype=out_dtype)
    u8 = torch.empty(n, ch, cw, 3, device=dev, dtype=torch.uint8) if want_u8 else None
    d.tmp, d.out, d.out_dtype, d.out_u8, d.rescale = _ptr(tmp), _ptr(out), dt(out), _ptr(u8), float(rescale)
    L.check(lib.dxa_image_preprocess(C.byref(d), _stream()), "dxa_image_preprocess")
    return (out, u8) if want_u8 else out


# ------------------------------------------------------------------------------------------------ fused DiT blocks
DIT_FUSED_MAX_ROWS, DIT_FUSED_MAX_TOKENS = 47, 32


def dit_blocks_supported(N: int, T1: int, H: int, heads: int, I: int) -> bool:
    return (N * T1 <= DIT_FUSED_MAX_ROWS and T1 <= DIT_FUSED_MAX_TOKENS and H == heads * 64 and H <= 1024 and I % 64 == 0)


def dit_blocks_fwd(h: torch.Tensor, weight_table: torch.Tensor, depth: int, N: int, T1: int, H: int, heads: int, I: int,
                   eps: float) -> torch.Tensor:
    """all DiT blocks of one denoising call in one launch; h [N*T1, H] fp32 is updated in place.  ``weight_table`` is an
    int64 device tensor of depth*8 raw fp32 pointers (qkv_w, qkv_b, proj_w, proj_b, fc1_w, fc1_b, fc2_w, fc2_b per block)"""
    assert h.dtype == torch.float32 and h.is_contiguous() and h.shape == (N * T1, H)
    assert weight_table.dtype == torch.int64 and weight_table.numel() == depth * 8 and weight_table.is_cuda
    nbytes = lib.dxa_dit_blocks_workspace(N * T1, H, I)
    ws = torch.empty(nbytes, device=h.device, dtype=torch.uint8)
    L.check(lib.dxa_dit_blocks_fwd(_ptr(h), _ptr(weight_table), depth, N, T1, H, heads, I, float(eps), _ptr(ws), nbytes,
                                   _stream()), "dxa_dit_blocks_fwd")
    return h


def dit_sample_fwd(x: torch.Tensor, z_emb: torch.Tensor, t_emb: torch.Tensor, pos: torch.Tensor, x_w: torch.Tensor, x_b: torch.Tensor,
                   final_w: torch.Tensor, final_b: torch.Tensor, coef: torch.Tensor, nb: int, use_cfg: bool, cfg_scale: float,
                   weight_table: torch.Tensor, depth: int, T1: int, H: int, heads: int, I: int, eps: float) -> torch.Tensor:
    """the whole DDIM sampler in one persistent launch (dxa_dit_sample_fwd); x [nb, T1-1, A] fp32 is updated in place"""
    steps, A = t_emb.shape[0], x.shape[-1]
    N = z_emb.shape[0]
    for t in (x, z_emb, t_emb, pos, x_w, x_b, final_w, final_b, coef):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.is_cuda
    assert x.shape == (nb, T1 - 1, A) and z_emb.shape == (N, H) and t_emb.shape == (steps, H) and coef.shape == (steps, 4)
    assert pos.shape == (T1, H) and x_w.shape == (H, A) and final_w.shape == (A, H)
    nbytes = lib.dxa_dit_sample_workspace(N * T1, H, I)
    ws = torch.empty(nbytes, device=x.device, dtype=torch.uint8)
    L.check(lib.dxa_dit_sample_fwd(_ptr(x), _ptr(z_emb), _ptr(t_emb), _ptr(pos), _ptr(x_w), _ptr(x_b), _ptr(final_w), _ptr(final_b),
                                   _ptr(coef), steps, A, nb, int(use_cfg), float(cfg_scale), _ptr(weight_table), depth, N, T1, H,
                                   heads, I, float(eps), _ptr(ws), nbytes, _stream()), "dxa_dit_sample_fwd")
    return x


def dit_bf16_pack(weight_table: torch.Tensor, depth: int, H: int, I: int, out=None, per: bool = False):
    """bf16 operand copy of the DiT blocks' matrices for dit_sample_bf16_fwd (dxa_dit_bf16_pack): returns (arena, table) — keep both
    alive; `table` is the int64 device tensor of depth*10 pointers the sampler takes.  Re-pack when the fp32 weights change
    (``out=(arena, table)`` re-packs in place).  ``per``: the blocks of MemVLA's DiT with perceptual attention (dxa_dit_bf16_pack_per:
    14 source pointers and 16 table entries per block)."""
    n_src, n_tab = (14, 16) if per else (8, 10)
    assert weight_table.dtype == torch.int64 and weight_table.numel() == depth * n_src and weight_table.is_cuda
    nbytes = (lib.dxa_dit_bf16_pack_per_bytes if per else lib.dxa_dit_bf16_pack_bytes)(depth, H, I)
    if out is not None:
        arena, table = out
        assert arena.numel() == nbytes and table.numel() == depth * n_tab
    else:
        arena = torch.empty(nbytes, device=weight_table.device, dtype=torch.uint8)
        table = torch.empty(depth * n_tab, device=weight_table.device, dtype=torch.int64)
    fn = lib.dxa_dit_bf16_pack_per if per else lib.dxa_dit_bf16_pack
    L.check(fn(_ptr(weight_table), depth, H, I, _ptr(arena), nbytes, _ptr(table), _stream()), "dxa_dit_bf16_pack")
    return arena, table


def dit_sample_bf16_supported(N: int, T1: int, H: int, heads: int, I: int, P: int = 0) -> bool:
    """``P``: perceptual keys per sample (MemVLA's DiT; 0 = the plain blocks)"""
    return (N * T1 <= 48 and T1 <= DIT_FUSED_MAX_TOKENS and H == heads * 64 and H <= 1024 and H % 64 == 0 and I % 64 == 0 and
            (P == 0 or (P % 64 == 0 and 64 <= P <= 256 and T1 <= 24)))


def dit_sample_bf16_fwd(x: torch.Tensor, z_emb: torch.Tensor, t_emb: torch.Tensor, pos: torch.Tensor, x_w: torch.Tensor,
                        x_b: torch.Tensor, final_w: torch.Tensor, final_b: torch.Tensor, coef: torch.Tensor, nb: int, use_cfg: bool,
                        cfg_scale: float, packed_table: torch.Tensor, depth: int, T1: int, H: int, heads: int, I: int,
                        eps: float, per_kv: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dit_sample_fwd with bf16 MFMA operands (dxa_dit_sample_bf16_fwd; `packed_table` from dit_bf16_pack): the sampler of a model
    served in bfloat16; x [nb, T1-1, A] fp32 is updated in place.  ``per_kv`` [depth, N, P, 2, H] fp32: MemVLA's perceptual attention
    (dxa_dit_sample_bf16_per_fwd; `packed_table` from dit_bf16_pack(per=True))."""
    steps, A = t_emb.shape[0], x.shape[-1]
    N = z_emb.shape[0]
    for t in (x, z_emb, t_emb, pos, x_w, x_b, final_w, final_b, coef):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.is_cuda
    assert x.shape == (nb, T1 - 1, A) and z_emb.shape == (N, H) and t_emb.shape == (steps, H) and coef.shape == (steps, 4)
    assert pos.shape == (T1, H) and x_w.shape == (H, A) and final_w.shape == (A, H)
    assert packed_table.dtype == torch.int64 and packed_table.numel() == depth * (10 if per_kv is None else 16) and packed_table.is_cuda
    nbytes = lib.dxa_dit_sample_bf16_workspace(N * T1, H, I)
    ws = torch.empty(nbytes, device=x.device, dtype=torch.uint8)
    if per_kv is not None:
        P_ = per_kv.shape[2]
        assert per_kv.dtype == torch.float32 and per_kv.is_contiguous() and per_kv.shape == (depth, N, P_, 2, H)
        L.check(lib.dxa_dit_sample_bf16_per_fwd(_ptr(x), _ptr(z_emb), _ptr(t_emb), _ptr(pos), _ptr(x_w), _ptr(x_b), _ptr(final_w),
                                                _ptr(final_b), _ptr(coef), steps, A, nb, int(use_cfg), float(cfg_scale),
                                                _ptr(packed_table), _ptr(per_kv), P_, depth, N, T1, H, heads, I, float(eps), _ptr(ws),
                                                nbytes, _stream()), "dxa_dit_sample_bf16_per_fwd")
        return x
    L.check(lib.dxa_dit_sample_bf16_fwd(_ptr(x), _ptr(z_emb), _ptr(t_emb), _ptr(pos), _ptr(x_w), _ptr(x_b), _ptr(final_w),
                                        _ptr(final_b), _ptr(coef), steps, A, nb, int(use_cfg), float(cfg_scale), _ptr(packed_table),
                                        depth, N, T1, H, heads, I, float(eps), _ptr(ws), nbytes, _stream()), "dxa_dit_sample_bf16_fwd")
    return x


def dit_blocks_timed_out(stream: Optional["torch.cuda.Stream"] = None) -> bool:
    """True if a fused DiT launch on `stream` (default: the current one) gave up at a device-wide barrier since the last
    call (its result is garbage and the request must be re-run unfused).  Synchronises the stream."""
    flag = C.c_int(0)
    L.check(lib.dxa_dit_blocks_status(stream.cuda_stream if stream is not None else _stream(), C.byref(flag)),
            "dxa_dit_blocks_status")
    return bool(flag.value)


# ------------------------------------------------------------------------------------------- persistent decode step
DECODE_FUSED_MAX_WIDTH = 32768      # csrc/decode_fused.hip ACT_MAX
DECODE_FUSED_MAX_KEYS = 16383       # ... T_MAX


def decode_step_supported(d: int, Hq: int, Hkv: int, D: int, F: int) -> bool:
    return (D in (64, 128, 256) and d % 8 == 0 and F % 8 == 0 and (Hq * D) % 8 == 0 and Hq % Hkv == 0 and
            max(d, F, Hq * D) <= DECODE_FUSED_MAX_WIDTH and Hq <= 256)


def decode_step(layer_table: torch.Tensor, x_in: torch.Tensor, out: torch.Tensor, final_w: torch.Tensor, cos_row: torch.Tensor,
                sin_row: torch.Tensor, ws: torch.Tensor, n_layers: int, d: int, Hq: int, Hkv: int, D: int, F: int, slot: int,
                kv_lo: int, max_len: int, eps: float) -> torch.Tensor:
    """one new token of one sequence through every decoder layer + final norm in ONE persistent launch (dxa_decode_step);
    `layer_table`: int64 device tensor of n_layers * 9 pointers (see include/dexbotic_amd.h); the caches are appended in place"""
    assert layer_table.dtype == torch.int64 and layer_table.numel() == n_layers * 9 and layer_table.is_cuda
    for t in (x_in, out, final_w):
        assert t.dtype == torch.bfloat16 and t.is_contiguous() and t.numel() == d and t.is_cuda
    for t in (cos_row, sin_row):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == D // 2 and t.is_cuda
    q = L.DecodeDesc()
    q.layers, q.x_in, q.out, q.final_norm_w = _ptr(layer_table), _ptr(x_in), _ptr(out), _ptr(final_w)
    q.cos_row, q.sin_row = _ptr(cos_row), _ptr(sin_row)
    q.workspace, q.workspace_bytes = _ptr(ws), ws.numel() * ws.element_size()
    q.n_layers, q.d, q.Hq, q.Hkv, q.D, q.F = n_layers, d, Hq, Hkv, D, F
    q.slot, q.kv_lo, q.max_len, q.eps = slot, kv_lo, max_len, float(eps)
    L.check(lib.dxa_decode_step(C.byref(q), _stream()), "dxa_decode_step")
    return out


def decode_step_workspace(d: int, Hq: int, Hkv: int, D: int, F: int, device) -> torch.Tensor:
    return torch.empty(lib.dxa_decode_step_workspace(d, Hq, Hkv, D, F), device=device, dtype=torch.uint8)


def decode_timed_out(stream: Optional["torch.cuda.Stream"] = None) -> bool:
    """True if a persistent decode launch on `stream` gave up at a device-wide barrier since the last call (its tokens are
    garbage).  Synchronises the stream."""
    flag = C.c_int(0)
    L.check(lib.dxa_decode_status(stream.cuda_stream if stream is not None else _stream(), C.byref(flag)), "dxa_decode_status")
    return bool(flag.value)
