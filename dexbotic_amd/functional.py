"""Block-level autograd Functions of the CogACT path.

Each Function is one transformer block (or one front-end / head stage): its forward is a fixed
sequence of libdexbotic_amd kernel launches, its backward the hand-scheduled reverse sequence.
Weight gradients are NOT returned to autograd: the dW GEMMs (TN layout, fp32 output) write — or
accumulate, for gradient accumulation — straight into the fp32 gradient arena of the ParamStore
(engine.py), which then notifies the data-parallel reducer that the block's bucket is complete.
An ``anchor`` parameter (any trainable tensor of the block) is passed through ``apply`` only so that
autograd schedules the backward even when the activations upstream do not require grad.

With 288 GB of HBM per MI355X every block keeps its activations by default; the reference's gradient
checkpointing (base_exp.py:245, trainer.py:101,120: HF checkpoints every decoder / encoder layer) is the
opt-in ``ParamStore.recompute`` (``model.gradient_checkpointing_enable()``): the transformer-layer Functions
then keep only their INPUT and re-run their forward launches at the top of their backward (same kernels, same
order: bit-identical gradients) — ~0.75 GB -> 33 MB kept per decoder layer at 16 x 287 tokens, for a 4th forward.
"""
from __future__ import annotations

import os

from dataclasses import dataclass
from typing import Optional, Tuple

import torch
from torch.autograd import Function

from . import _lib as L
from . import kernels as K
from .engine import ParamStore


def _names(x) -> Tuple[str, ...]:
    return (x,) if isinstance(x, str) else tuple(x)


_OUTER_GRAD = [True]     # grad mode at the apply() call site (inside Function.forward it always reads False)


class _StoreFn(Function):
    """autograd Function that writes parameter gradients into the ParamStore arenas.  Its backward runs the fp32 products in the
    mode its forward ran in (kernels.F32_GEMM_MODE: exact fp32 MFMA or the split-bf16 product): a training loop this package does
    not manage calls ``loss.backward()`` long after the model's forward — and whatever scoped the mode — has returned."""

    def __init_subclass__(cls, **kw):
        super().__init_subclass__(**kw)
        fwd, bwd = cls.__dict__.get("forward"), cls.__dict__.get("backward")
        if fwd is None or bwd is None:
            return
        f0 = fwd.__func__ if isinstance(fwd, staticmethod) else fwd
        b0 = bwd.__func__ if isinstance(bwd, staticmethod) else bwd

        def forward(ctx, *args, **kwargs):
            ctx._f32_mode = K.F32_GEMM_MODE
            return f0(ctx, *args, **kwargs)

        def backward(ctx, *grads):
            mode = getattr(ctx, "_f32_mode", None)
            if mode is None or mode == K.F32_GEMM_MODE:
                return b0(ctx, *grads)
            with K.f32_gemm_mode(mode):
                return b0(ctx, *grads)
        forward.__doc__, backward.__doc__ = f0.__doc__, b0.__doc__
        cls.forward, cls.backward = staticmethod(forward), staticmethod(backward)

    @classmethod
    def apply(cls, *args, **kwargs):
        _OUTER_GRAD[0] = torch.is_grad_enabled()
        return super().apply(*args, **kwargs)


def _use(ctx, st: ParamStore, *groups) -> None:
    """forward-side half of the bucket accounting (ParamStore.note_use): this Function's backward will write each of
    these gradient slots once.  Only when a backward can actually run (grad mode on, some input requires grad)."""
    if not (_OUTER_GRAD[0] and any(ctx.needs_input_grad)):
        return
    for g in groups:
        if g is None:
            continue
        st.note_use(*[n for n in _names(g) if n is not None])


def _recompute(ctx, st: ParamStore) -> bool:
    """activation recompute for this call: opted in on the store and a backward can actually run"""
    return bool(getattr(st, "recompute", False)) and _OUTER_GRAD[0] and any(ctx.needs_input_grad)


def _f32_nt(dy2d: torch.Tensor) -> bool:
    """fp32 products of the action head (>= 128 rows): the tiled kernel's k-contiguous (NT) staging is ~1.5-2.5x faster
    than its k-strided NN / TN staging (scripts/gemm_f32_bench.py), so transposing a small operand first pays"""
    return dy2d.dtype == torch.float32 and dy2d.shape[0] >= 128 and dy2d.shape[1] >= 128


def _dx(st: ParamStore, names, wshape, dy2d: torch.Tensor, **kw) -> torch.Tensor:
    """dX = dY W for W = fused view over `names` of shape wshape = (out, in).  bf16: an NN product on the ping-pong MFMA
    kernel that stages W as it lies in memory (k-strided operand, ds_read_b64_tr_b16 fragments) — no W^T copy exists."""
    names = _names(names)
    W = st.w(*names, shape=tuple(wshape))
    if _f32_nt(dy2d):
        return K.mm_nn_f32(dy2d, W, **kw)
    return K.mm_nn(dy2d, W, **kw)


def _wgrad(st: ParamStore, names, dy2d: torch.Tensor, x2d: torch.Tensor, shape) -> None:
    """dW (+)= dy^T x into the gradient arena (fused view over `names`).  bf16: a TN product on the ping-pong MFMA kernel:
    both activations are staged as they lie ([tokens, features], the contraction runs over the rows) — no transposed
    copies, any token count.  A parameter with several consumers in this forward (``ParamStore.note_use``) collects their
    (dy, x) pairs and writes dW with ONE product over all rows when the last consumer's backward arrives
    (``ParamStore.defer_wgrad``)."""
    names = _names(names)
    if not all(st.trainable(n) for n in names):
        if any(st.trainable(n) for n in names):
            raise L.DxaError(f"fused parameters {names} must be frozen/unfrozen together")
        return
    key = tuple(names)
    pending = st._uses.get(names[0], 0)                  # consumers whose backward has not run yet, this one included
    if st.defer_wgrad and (pending > 1 or key in st._wg_stash):
        ent = st._wg_stash.get(key)
        if ent is None:
            ent = st._wg_stash[key] = {"acc0": st.accum_flag(*names), "dy": [], "x": [], "shape": shape}
        ent["dy"].append(dy2d)
        ent["x"].append(x2d)
        if pending > 1:
            st.mark_written(*names)                      # counts this consumer down; the bucket waits for the last one
            return
        _wgrad_flush(st, key)
        return
    if st.accum_merge and dy2d.dtype == torch.bfloat16 and x2d.dtype == torch.bfloat16 and dy2d.is_contiguous() and x2d.is_contiguous():
        # gradient accumulation over TWO micro-batches (the reference recipe): the first micro-batch keeps its (dY, X) and
        # writes nothing; the last one contracts both pairs in ONE product — dW is written once instead of written, read back
        # and written again (ParamStore.accum_merge; the trainer arms it for grad_accum == 2)
        if not st.last_micro and key not in st._accum_stash:
            st._accum_stash[key] = (dy2d, x2d, shape)
            return
        held = st._accum_stash.pop(key, None)
        if held is not None:
            if st.last_micro and held[0].shape[1] == dy2d.shape[1] and held[1].shape[1] == x2d.shape[1]:
                _wgrad_now(st, names, dy2d, x2d, shape, st.accum_flag(*names), seg2=held[:2])
                return
            _wgrad_now(st, names, held[0], held[1], held[2], st.accum_flag(*names))      # (no partner: written on its own)
    _wgrad_now(st, names, dy2d, x2d, shape, st.accum_flag(*names))


def flush_accum(st: ParamStore) -> None:
    """(dY, X) pairs of the first micro-batch whose parameter the last micro-batch did not use: their dW is written now"""
    for key in list(st._accum_stash):
        dy, x, shape = st._accum_stash.pop(key)
        _wgrad_now(st, key, dy, x, shape, st.accum_flag(*key))


def _wgrad_flush(st: ParamStore, key: tuple) -> None:
    ent = st._wg_stash.pop(key)
    dy = ent["dy"][0] if len(ent["dy"]) == 1 else torch.cat(ent["dy"], dim=0)
    x = ent["x"][0] if len(ent["x"]) == 1 else torch.cat(ent["x"], dim=0)
    _wgrad_now(st, key, dy.contiguous(), x.contiguous(), ent["shape"], ent["acc0"], first_write=not ent["acc0"])


def _wgrad_now(st: ParamStore, names, dy2d: torch.Tensor, x2d: torch.Tensor, shape, accumulate: bool,
               first_write: Optional[bool] = None, seg2=None) -> None:
    side = st.wgrad_stream
    if side is not None and st.wgrad_stream_f32_only and dy2d.dtype != torch.float32:
        side = None                                # (only the fp32 action head's products leave the compute stream)
    if side is None:
        _wgrad_product(st, names, dy2d, x2d, shape, accumulate, first_write, seg2)
    else:
        # the product runs on the side stream, ordered after everything enqueued on the compute stream so far (its operands);
        # all dW products share that stream, so successive writes of one slot stay ordered; readers join (ParamStore.join_wgrad)
        side.wait_stream(torch.cuda.current_stream(st.device))
        with torch.cuda.stream(side):
            _wgrad_product(st, names, dy2d, x2d, shape, accumulate, first_write, seg2)
        for t in (dy2d, x2d) + (tuple(seg2) if seg2 is not None else ()):
            t.record_stream(side)                  # the allocator must not hand the operands out again before the product has read them
        st._wgrad_pending = True
    st.mark_written(*names)


def _wgrad_product(st: ParamStore, names, dy2d: torch.Tensor, x2d: torch.Tensor, shape, accumulate: bool,
                   first_write: Optional[bool], seg2) -> None:
    kw2 = {} if seg2 is None else {"a2": seg2[0], "b2": seg2[1]}
    if st.bf16_grads and st.gradc is not None and dy2d.dtype == torch.bfloat16 and x2d.dtype == torch.bfloat16:
        # bf16 gradient arena: the product writes bf16 only (ParamStore.bf16_grads); the slots count as already copied
        out = st.gc(*names, shape=shape)
        ssq = st.sumsq_out(names, out.shape[0], out.shape[1], first_write) if out.dim() == 2 else None
        K.mm_tn(dy2d, x2d, out=out, accumulate=accumulate, sumsq=ssq, **kw2)
        st._mirrored.update(names)
        return
    out = st.g(*names, shape=shape)
    # bf16 data parallelism: the product's epilogue also writes the bf16 communication copy of this gradient
    mirror = st.mirror_out(*names, shape=shape)
    # single-GPU clip: ... and this gradient's share of sum(g^2) (bf16 gradient arena: the norm is taken over the bf16 copy
    # AdamW reads, so an fp32 product's share is read back from that copy instead)
    ssq = st.sumsq_out(names, out.shape[0], out.shape[1], first_write) if out.dim() == 2 and not st.bf16_grads else None
    if _f32_nt(dy2d) and x2d.dtype == torch.float32:
        K.mm_tn_f32(dy2d, x2d, out=out, accumulate=accumulate, mirror=mirror, sumsq=ssq)
    else:
        K.mm_tn(dy2d, x2d, out=out, accumulate=accumulate, mirror=mirror, sumsq=ssq, **kw2)


_SKIP_BGRAD = os.environ.get("DXA_TUNE_SKIP_BGRAD") == "1"
_NO_DEFER_BGRAD = os.environ.get("DXA_NO_DEFER_BGRAD") == "1"      # A/B: one column sum per consumer of a bias (rounds 1 - 6a)


def _bgrad(st: ParamStore, names, dy2d: torch.Tensor) -> None:
    names = _names(names)
    if not all(st.trainable(n) for n in names):
        return
    n = sum(st.slots[nm].numel for nm in names)
    if _SKIP_BGRAD:            # tuning only (WRONG gradients): what the step would cost if the bias column sums were free
        st.mark_written(*names)
        return
    # a bias applied k times in one forward (MemVLA's per-sample retrieval blocks: 16 samples x 2 roles x 11 linears): the k dY are
    # collected like the weight's (dY, X) pairs (_wgrad) and ONE column sum over all their rows writes db once — k - 1 launches of the
    # backward's host-bound stretch less per bias (profiles/r06_host_uploads.txt: the bank's backward is issued at 21 us a launch)
    key = tuple(names)
    pending = st._uses.get(names[0], 0)
    acc0 = None
    if st.defer_wgrad and not _NO_DEFER_BGRAD and (pending > 1 or key in st._bg_stash):
        ent = st._bg_stash.get(key)
        if ent is None:
            ent = st._bg_stash[key] = {"acc0": st.accum_flag(*names), "dy": []}
        ent["dy"].append(dy2d)
        if pending > 1:
            st.mark_written(*names)                      # counts this consumer down; the bucket waits for the last one
            return
        ent = st._bg_stash.pop(key)
        dy2d = ent["dy"][0] if len(ent["dy"]) == 1 else torch.cat(ent["dy"], dim=0)
        acc0 = ent["acc0"]
    _bgrad_now(st, names, n, dy2d, st.accum_flag(*names) if acc0 is None else acc0)


def _bgrad_flush(st: ParamStore, key: tuple) -> None:
    ent = st._bg_stash.pop(key)
    dy = ent["dy"][0] if len(ent["dy"]) == 1 else torch.cat(ent["dy"], dim=0)
    _bgrad_now(st, key, sum(st.slots[nm].numel for nm in key), dy, ent["acc0"])


def _bgrad_now(st: ParamStore, names, n: int, dy2d: torch.Tensor, accumulate: bool) -> None:
    side = st.wgrad_stream if st.bgrad_on_side else None
    if side is None:
        K.colsum(dy2d, out=st.g(*names, shape=(n,)), accumulate=accumulate)
    else:
        side.wait_stream(torch.cuda.current_stream(st.device))
        with torch.cuda.stream(side):
            K.colsum(dy2d, out=st.g(*names, shape=(n,)), accumulate=accumulate)
        dy2d.record_stream(side)
        st._wgrad_pending = True
    st.mark_written(*names)


def _vgrad(st: ParamStore, name: str, val: torch.Tensor) -> None:
    """small fp32 gradient vector/matrix (already reduced) into the arena"""
    if not st.trainable(name):
        return
    g = st.g(name)
    flat = g.view(-1)
    v = val.reshape(-1)
    if st.accum_flag(name):
        K.add(flat, v.contiguous(), out=flat)
    else:
        K.cast(v.contiguous(), torch.float32, out=flat)
    st.mark_written(name)


# ------------------------------------------------------------------------------------------- Qwen2 layer
@dataclass
class Qwen2LayerSpec:
    ln1: str
    qkv_w: Tuple[str, str, str]
    qkv_b: Tuple[str, str, str]
    o_w: str
    ln2: str
    gu_w: Tuple[str, str]
    down_w: str
    B: int = 0
    S: int = 0
    Hq: int = 0
    Hkv: int = 0
    D: int = 0
    d: int = 0
    F: int = 0
    eps: float = 1e-6
    grad_mode: bool = True     # grad mode of the caller of the current forward (Qwen2Backbone.forward records it)


class Qwen2LayerFn(_StoreFn):
    """HF Qwen2DecoderLayer (HF:qwen2/modeling_qwen2.py:258-300) called from cogact_arch.py:97-106:
    x + o_proj(attn(rope(qkv(rmsnorm(x))))) ; then + down(silu(gate)*up) of rmsnorm."""

    @staticmethod
    def _run(st: ParamStore, sp: Qwen2LayerSpec, x, cos_t, sin_t, kv_start, kv_end, keep: bool = True):
        """the layer's forward launches -> (y, what the backward reads besides x).  ``keep`` = False (no gradient will be asked
        for: serving): the gated MLP's pre-activations are not stored and SiLU * up runs in the gate / up product's epilogue"""
        B, S, Hq, Hkv, D, d, F_ = sp.B, sp.S, sp.Hq, sp.Hkv, sp.D, sp.d, sp.F
        M = B * S
        nq = (Hq + 2 * Hkv) * D
        h1, rstd1 = K.rmsnorm_fwd(x, st.w(sp.ln1), sp.eps)
        qkv = K.mm_nt(h1, st.w(*sp.qkv_w, shape=(nq, d)), bias=st.w(*sp.qkv_b, shape=(nq,)))
        q, k, v = K.rope_split(qkv, cos_t, sin_t, None, B, S, Hq, Hkv, D)
        del qkv
        o = torch.empty((B, S, Hq, D), device=x.device, dtype=x.dtype)
        lse = K.attn_fwd(q, k, v, o.permute(0, 2, 1, 3), causal=True, scale=D ** -0.5, kv_start=kv_start, kv_end=kv_end)
        x2 = K.mm_nt(o.view(M, Hq * D), st.w(sp.o_w), residual=x)
        h2, rstd2 = K.rmsnorm_fwd(x2, st.w(sp.ln2), sp.eps)
        w_gu = st.w(*sp.gu_w, shape=(2 * F_, d))
        if K.swiglu_gemm_supported(h2, w_gu, keep_pre=keep):          # SiLU(gate) * up in the product's own epilogue
            a, gu = K.mm_nt_swiglu(h2, w_gu, keep_pre=keep)
        else:
            gu = K.mm_nt(h2, w_gu)
            a = K.swiglu_fwd(gu)
        y = K.mm_nt(a, st.w(sp.down_w), residual=x2)
        return y, (rstd1, h1, q, k, v, o, lse, x2, rstd2, h2, gu, a)

    @staticmethod
    def forward(ctx, x, anchor, st: ParamStore, sp: Qwen2LayerSpec, cos_t, sin_t, kv_start, kv_end):
        # will a backward come?  (needs_input_grad reports requires_grad of the inputs whatever the grad mode of the caller: a served
        #  model keeps its parameters trainable, so the caller's grad mode — recorded by Qwen2Backbone.forward — decides too)
        need = bool(ctx.needs_input_grad[0] or ctx.needs_input_grad[1]) and getattr(sp, "grad_mode", True)
        y, saved = Qwen2LayerFn._run(st, sp, x, cos_t, sin_t, kv_start, kv_end, keep=need)
        if not need:
            return y
        ctx.st, ctx.sp = st, sp
        _use(ctx, st, sp.ln1, sp.qkv_w, sp.qkv_b, sp.o_w, sp.ln2, sp.gu_w, sp.down_w)
        ctx.aux = (cos_t, sin_t, kv_start, kv_end)
        ctx.recompute = _recompute(ctx, st)
        if ctx.recompute:
            ctx.save_for_backward(x)              # gradient checkpointing: the input only, the rest is re-run in backward
        else:
            ctx.save_for_backward(x, *saved)
        return y

    @staticmethod
    def backward(ctx, dy):
        st, sp = ctx.st, ctx.sp
        cos_t, sin_t, kv_start, kv_end = ctx.aux
        if ctx.recompute:
            (x,) = ctx.saved_tensors
            rstd1, h1, q, k, v, o, lse, x2, rstd2, h2, gu, a = Qwen2LayerFn._run(st, sp, x, cos_t, sin_t, kv_start, kv_end)[1]
        else:
            x, rstd1, h1, q, k, v, o, lse, x2, rstd2, h2, gu, a = ctx.saved_tensors
        B, S, Hq, Hkv, D, d, F_ = sp.B, sp.S, sp.Hq, sp.Hkv, sp.D, sp.d, sp.F
        M = B * S
        nq = (Hq + 2 * Hkv) * D
        dy = dy.contiguous()
        # ---- MLP
        da = _dx(st, sp.down_w, (d, F_), dy)
        _wgrad(st, sp.down_w, dy, a, (d, F_))
        dgu = K.swiglu_bwd(gu, da)
        del da
        dh2 = _dx(st, sp.gu_w, (2 * F_, d), dgu)
        _wgrad(st, sp.gu_w, dgu, h2, (2 * F_, d))
        del dgu
        tr2 = st.trainable(sp.ln2)
        dx2, _ = K.rmsnorm_bwd(dh2, x2, st.w(sp.ln2), rstd2, dw_out=st.g(sp.ln2) if tr2 else None,
                               accumulate=st.accum_flag(sp.ln2), want_dw=tr2, residual=dy)     # dy + norm backward
        if tr2:
            st.mark_written(sp.ln2)
        del dh2
        # ---- attention: dO written head-major by a (b, h)-batched NN GEMM so the GQA group folds into
        #      the rows of the dK/dV GEMMs (attention.hip)
        if dx2.dtype == torch.bfloat16:
            do = K.permute_bshd(_dx(st, sp.o_w, (d, Hq * D), dx2), B, S, Hq, D, True)
        else:
            wo = st.w(sp.o_w)                                    # [d, Hq*D]
            do = torch.empty((B, Hq, S, D), device=x.device, dtype=x.dtype)
            K.gemm(L.NN, dx2, wo, S, D, d, d, Hq * D, do, D, nb=(B, Hq, 1),
                   sA=(S * d, 0, 0), sB=(0, D, 0), sC=(Hq * S * D, S * D, 0))
        _wgrad(st, sp.o_w, dx2, o.view(M, Hq * D), (d, Hq * D))
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        K.attn_bwd(q, k, v, o.permute(0, 2, 1, 3), lse, do, dq, dk, dv, causal=True, scale=D ** -0.5,
                   kv_start=kv_start, kv_end=kv_end)
        dqkv = K.rope_merge(dq, dk, dv, cos_t, sin_t, None, B, S, Hq, Hkv, D)
        del dq, dk, dv, do
        dh1 = _dx(st, sp.qkv_w, (nq, d), dqkv)
        _wgrad(st, sp.qkv_w, dqkv, h1, (nq, d))
        _bgrad(st, sp.qkv_b, dqkv)
        del dqkv
        tr1 = st.trainable(sp.ln1)
        dx, _ = K.rmsnorm_bwd(dh1, x, st.w(sp.ln1), rstd1, dw_out=st.g(sp.ln1) if tr1 else None,
                              accumulate=st.accum_flag(sp.ln1), want_dw=tr1, residual=dx2)
        if tr1:
            st.mark_written(sp.ln1)
        return dx, None, None, None, None, None, None, None


# ------------------------------------------------------------------- ViT-style block (CLIP layer, DiT block)
@dataclass
class VitBlockSpec:
    ln1_w: Optional[str]
    ln1_b: Optional[str]
    qkv_w: Tuple[str, ...]      # 3 adjacent tensors (CLIP) or the single fused one (timm)
    qkv_b: Tuple[str, ...]
    out_w: str
    out_b: str
    ln2_w: Optional[str]
    ln2_b: Optional[str]
    fc1_w: str
    fc1_b: str
    fc2_w: str
    fc2_b: str
    act: int
    eps: float
    N: int = 0      # sequences
    T: int = 0      # tokens per sequence
    H: int = 0
    D: int = 0
    I: int = 0      # mlp width
    ckpt: bool = True   # recomputed under ParamStore.recompute (HF checkpoints its encoder layers; the DiT head's blocks are
                        # plain nn.Modules in the reference and keep their activations: False there)


def _padded_head_dim(D: int, dtype) -> int:
    """head width the MFMA attention kernels run at for a model head_dim D (bf16 only; fp32 uses the generic kernels)"""
    # 72 (SigLIP-So400m) runs natively on the 128-wide tiles since round 5: only its 72 real columns are loaded / stored
    # (DXA_ATTN_PAD72=1: the round-3/4 zero-padded copies, kept for the A/B of profiles/r05_pi0_hd72.txt)
    native = (64, 128, 256) if os.environ.get("DXA_ATTN_PAD72") == "1" else (64, 72, 128, 256)
    if dtype != torch.bfloat16 or D in native or D > 256 or D % 8 != 0:
        return D
    return 64 if D < 64 else (128 if D < 128 else 256)


class VitBlockFn(_StoreFn):
    """Pre-LN block: x + out(attn(qkv(LN(x)))) ; + fc2(act(fc1(LN(.)))).
    CLIP encoder layer (HF:clip/modeling_clip.py:259-384; affine LN eps 1e-5, quick_gelu) and DiTBlock
    (cogact/action_model/dit.py:137-162 with timm Attention/Mlp; LN without affine eps 1e-6, tanh-GELU)."""

    @staticmethod
    def _run(st: ParamStore, sp: VitBlockSpec, x):
        """the block's forward launches on x [M, C] -> (y [M, C], what the backward reads besides x)"""
        N, T, H, D, I = sp.N, sp.T, sp.H, sp.D, sp.I
        C_ = H * D
        M = N * T
        w = lambda n: st.w(n) if n is not None else None
        h1, mean1, rstd1 = K.layernorm_fwd(x, w(sp.ln1_w), w(sp.ln1_b), sp.eps)
        qkv = K.mm_nt(h1, st.w(*sp.qkv_w, shape=(3 * C_, C_)), bias=st.w(*sp.qkv_b, shape=(3 * C_,)))
        Dp = _padded_head_dim(D, x.dtype)
        if Dp != D:
            # head_dim the MFMA attention kernels do not take (SigLIP-So400m: 72): run them at the next supported
            # width on zero-padded heads — q.k is unchanged by zero columns, the padded output columns are dropped
            qkv = K.copy2d(qkv.view(M * 3 * H, D), torch.empty((M * 3 * H, Dp), device=x.device, dtype=x.dtype), D, Dp)
        q5 = qkv.view(N, T, 3, H, Dp)
        q, k, v = (q5[:, :, i].permute(0, 2, 1, 3) for i in range(3))
        o = torch.empty((N, T, H, Dp), device=x.device, dtype=x.dtype)
        lse = K.attn_fwd(q, k, v, o.permute(0, 2, 1, 3), causal=False, scale=D ** -0.5)
        o_in = o if Dp == D else K.copy2d(o.view(M * H, Dp), torch.empty((M * H, D), device=x.device, dtype=x.dtype), D, D)
        x2 = K.mm_nt(o_in.view(M, C_), st.w(sp.out_w), bias=st.w(sp.out_b), residual=x)
        h2, mean2, rstd2 = K.layernorm_fwd(x2, w(sp.ln2_w), w(sp.ln2_b), sp.eps)
        pre = torch.empty((M, I), device=x.device, dtype=x.dtype)
        a = K.mm_nt(h2, st.w(sp.fc1_w), bias=st.w(sp.fc1_b), act=sp.act, aux_out=pre)
        y = K.mm_nt(a, st.w(sp.fc2_w), bias=st.w(sp.fc2_b), residual=x2)
        return y, (mean1, rstd1, h1, qkv, o, lse, x2, mean2, rstd2, h2, pre, a)

    @staticmethod
    def forward(ctx, x, anchor, st: ParamStore, sp: VitBlockSpec):
        N, T, C_ = sp.N, sp.T, sp.H * sp.D
        x = x.reshape(N * T, C_)
        y, saved = VitBlockFn._run(st, sp, x)
        ctx.st, ctx.sp = st, sp
        _use(ctx, st, sp.ln1_w, sp.ln1_b, sp.qkv_w, sp.qkv_b, sp.out_w, sp.out_b, sp.ln2_w, sp.ln2_b, sp.fc1_w, sp.fc1_b,
             sp.fc2_w, sp.fc2_b)
        ctx.recompute = sp.ckpt and _recompute(ctx, st)
        if ctx.recompute:
            ctx.save_for_backward(x)
        else:
            ctx.save_for_backward(x, *saved)
        return y.view(N, T, C_)

    @staticmethod
    def backward(ctx, dy):
        st, sp = ctx.st, ctx.sp
        if ctx.recompute:
            (x,) = ctx.saved_tensors
            mean1, rstd1, h1, qkv, o, lse, x2, mean2, rstd2, h2, pre, a = VitBlockFn._run(st, sp, x)[1]
        else:
            x, mean1, rstd1, h1, qkv, o, lse, x2, mean2, rstd2, h2, pre, a = ctx.saved_tensors
        N, T, H, D, I = sp.N, sp.T, sp.H, sp.D, sp.I
        C_ = H * D
        M = N * T
        w = lambda n: st.w(n) if n is not None else None
        dy = dy.reshape(M, C_).contiguous()
        # ---- MLP (activation gradient fused into the dX GEMM epilogue)
        dpre = _dx(st, sp.fc2_w, (C_, I), dy, mulgrad=pre, act=sp.act)
        _wgrad(st, sp.fc2_w, dy, a, (C_, I))
        _bgrad(st, sp.fc2_b, dy)
        dh2 = _dx(st, sp.fc1_w, (I, C_), dpre)
        _wgrad(st, sp.fc1_w, dpre, h2, (I, C_))
        _bgrad(st, sp.fc1_b, dpre)
        del dpre
        dx2 = _ln_bwd(st, dh2, x2, sp.ln2_w, sp.ln2_b, mean2, rstd2, residual=dy.reshape(x2.shape))
        del dh2
        # ---- attention
        do = _dx(st, sp.out_w, (C_, C_), dx2)                   # [M, C] token-major
        Dp = o.shape[-1]                                          # padded head width of the forward (== D normally)
        o_in = o if Dp == D else K.copy2d(o.view(M * H, Dp), torch.empty((M * H, D), device=o.device, dtype=o.dtype), D, D)
        _wgrad(st, sp.out_w, dx2, o_in.view(M, C_), (C_, C_))
        _bgrad(st, sp.out_b, dx2)
        if Dp != D:
            do = K.copy2d(do.view(M * H, D), torch.empty((M * H, Dp), device=o.device, dtype=o.dtype), D, Dp)
        dqkv = torch.empty_like(qkv)
        q5, d5 = qkv.view(N, T, 3, H, Dp), dqkv.view(N, T, 3, H, Dp)
        q, k, v = (q5[:, :, i].permute(0, 2, 1, 3) for i in range(3))
        dq, dk, dv = (d5[:, :, i].permute(0, 2, 1, 3) for i in range(3))
        K.attn_bwd(q, k, v, o.permute(0, 2, 1, 3), lse, do.view(N, T, H, Dp).permute(0, 2, 1, 3), dq, dk, dv,
                   causal=False, scale=D ** -0.5)
        if Dp != D:
            dqkv = K.copy2d(dqkv.view(M * 3 * H, Dp), torch.empty((M * 3 * H, D), device=o.device, dtype=o.dtype), D, D)
            dqkv = dqkv.view(M, 3 * C_)
        dh1 = _dx(st, sp.qkv_w, (3 * C_, C_), dqkv)
        _wgrad(st, sp.qkv_w, dqkv, h1, (3 * C_, C_))
        _bgrad(st, sp.qkv_b, dqkv)
        del dqkv
        dx = _ln_bwd(st, dh1, x, sp.ln1_w, sp.ln1_b, mean1, rstd1, residual=dx2.reshape(x.shape))
        return dx.view(N, T, C_), None, None, None


def _ln_bwd(st: ParamStore, dy, x, wn: Optional[str], bn: Optional[str], mean, rstd, residual=None) -> torch.Tensor:
    """LayerNorm backward; `residual` (gradient of the skip connection around the sub-block) is added inside the kernel"""
    if wn is None:
        dx, _, _ = K.layernorm_bwd(dy, x, None, mean, rstd, residual=residual)
        return dx
    tr = st.trainable(wn)
    if tr and bn is not None and st.defer_wgrad and not _NO_DEFER_BGRAD and (st._uses.get(wn, 0) > 1 or (wn, bn) in st._bg_stash) and \
            st.slots[bn].offset == st.slots[wn].offset + st.slots[wn].numel:
        # an affine LayerNorm applied k times per forward (MemVLA's retrieval blocks): the kernel's per-row-block partial sums
        # [blocks, dw | db] of the k calls are folded by ONE column sum when the last one arrives (_bgrad's stash) instead of one per call
        dx, part = K.layernorm_bwd(dy, x, st.w(wn), mean, rstd, residual=residual, return_part=True)
        _bgrad(st, (wn, bn), part)
        return dx
    dx, _, _ = K.layernorm_bwd(dy, x, st.w(wn), mean, rstd, dw_out=st.g(wn) if tr else None,
                               db_out=st.g(bn) if tr else None, accumulate=st.accum_flag(wn), want_dw=tr, residual=residual)
    if tr:
        st.mark_written(wn, bn)
    return dx


# --------------------------------------------------------------------------------------- generic pieces
class LinearFn(_StoreFn):
    """y = act(x W^T + b) (+ residual).  nn.Linear call sites of the path that are not inside a block:
    patch embedding (HF:clip/modeling_clip.py:149-155 as a GEMM over im2col rows), DiT embedders and final
    layer (dit.py:22-64,106-135,165-178)."""

    @staticmethod
    def forward(ctx, x, anchor, st: ParamStore, wn: str, bn: Optional[str], act: int, wshape):
        x2 = x.reshape(-1, x.shape[-1])
        if not x2.is_contiguous() and not (x2.stride(1) == 1):
            x2 = x2.contiguous()
        W = st.w(wn, shape=wshape)
        pre = None
        if act != L.ACT_NONE:
            pre = torch.empty((x2.shape[0], W.shape[0]), device=x.device, dtype=x.dtype)
        y = K.mm_nt(x2[:, :W.shape[1]] if x2.shape[1] != W.shape[1] else x2, W,
                    bias=st.w(bn) if bn else None, act=act, aux_out=pre)
        ctx.st, ctx.wn, ctx.bn, ctx.act, ctx.wshape, ctx.xshape = st, wn, bn, act, wshape, x.shape
        _use(ctx, st, wn, bn)
        ctx.save_for_backward(x2, pre if pre is not None else x2.new_empty(0))
        return y.view(*x.shape[:-1], W.shape[0])

    @staticmethod
    def backward(ctx, dy):
        st = ctx.st
        x2, pre = ctx.saved_tensors
        W = st.w(ctx.wn, shape=ctx.wshape)
        dy2 = dy.reshape(-1, W.shape[0]).contiguous()
        if ctx.act != L.ACT_NONE:
            dy2 = K.act_bwd(pre, dy2, ctx.act)
        xk = x2[:, :W.shape[1]] if x2.shape[1] != W.shape[1] else x2
        _wgrad(st, ctx.wn, dy2, xk, W.shape)
        if ctx.bn:
            _bgrad(st, ctx.bn, dy2)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _dx(st, ctx.wn, W.shape, dy2)
            if x2.shape[1] != W.shape[1]:
                raise L.DxaError("LinearFn: padded input cannot receive a gradient")
            dx = dx.view(ctx.xshape)
        return dx, None, None, None, None, None, None


class PackedInProjFn(_StoreFn):
    """nn.MultiheadAttention's packed in-projection for CROSS attention (torch F._in_projection_packed with k is v):
    q = x W[:E]^T + b[:E] from the query rows, [k | v] = y W[E:]^T + b[E:] from the key/value rows — the perceptual
    attention of the MemVLA DiT block (memvla/action_model/dit.py:158-185).  The two products use disjoint row ranges of the
    one packed parameter: its gradient is written range by range (dW[:E] = dq^T x, dW[E:] = dkv^T y) inside ONE Function, so
    the slot is announced and marked written once.  (Applying the whole 3E-row projection to both inputs and slicing, as
    round 2 did, computed a third of the key/value projection — 16384 perceptual rows per block at the MemVLA batch — for
    nothing.)"""

    @staticmethod
    def forward(ctx, x, y, anchor, st: ParamStore, wn: str, bn: str, E: int):
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        y2 = y.reshape(-1, y.shape[-1]).contiguous()
        W, b = st.w(wn), st.w(bn)
        q = K.mm_nt(x2, W[:E], bias=b[:E])
        kv = K.mm_nt(y2, W[E:], bias=b[E:])
        ctx.st, ctx.wn, ctx.bn, ctx.E = st, wn, bn, E
        _use(ctx, st, wn, bn)
        ctx.save_for_backward(x2, y2)
        return q, kv

    @staticmethod
    def backward(ctx, dq, dkv):
        st, wn, bn, E = ctx.st, ctx.wn, ctx.bn, ctx.E
        x2, y2 = ctx.saved_tensors
        W = st.w(wn)
        dq, dkv = dq.contiguous(), dkv.contiguous()
        dx = dy = None
        nt = _f32_nt(dq)
        if ctx.needs_input_grad[0]:
            dx = K.mm_nn_f32(dq, W[:E]) if nt else K.mm_nn(dq, W[:E])
        if ctx.needs_input_grad[1]:
            dy = K.mm_nn_f32(dkv, W[E:]) if _f32_nt(dkv) else K.mm_nn(dkv, W[E:])
        if st.trainable(wn):
            acc = st.accum_flag(wn)
            g, gb = st.g(wn), st.g(bn)
            for lo, hi, d2, in2 in ((0, E, dq, x2), (E, W.shape[0], dkv, y2)):
                if _f32_nt(d2) and in2.dtype == torch.float32:
                    K.mm_tn_f32(d2, in2, out=g[lo:hi], accumulate=acc)
                else:
                    K.mm_tn(d2, in2, out=g[lo:hi], accumulate=acc)
                K.colsum(d2, out=gb[lo:hi], accumulate=acc)
            st.mark_written(wn, bn)
        return dx, dy, None, None, None, None, None


class MlpFn(_StoreFn):
    """y = (act(x W1^T + b1)) W2^T + b2 : mm_projector mlp2x_gelu (mm_projector/builder.py:71-79) and the
    TimestepEmbedder MLP (dit.py:27-31)."""

    @staticmethod
    def forward(ctx, x, anchor, st: ParamStore, w1, b1, w2, b2, act: int):
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        W1, W2 = st.w(w1), st.w(w2)
        pre = torch.empty((x2.shape[0], W1.shape[0]), device=x.device, dtype=x.dtype)
        a = K.mm_nt(x2, W1, bias=st.w(b1), act=act, aux_out=pre)
        y = K.mm_nt(a, W2, bias=st.w(b2))
        ctx.st, ctx.names, ctx.act, ctx.xshape = st, (w1, b1, w2, b2), act, x.shape
        _use(ctx, st, w1, b1, w2, b2)
        ctx.save_for_backward(x2, pre, a)
        return y.view(*x.shape[:-1], W2.shape[0])

    @staticmethod
    def backward(ctx, dy):
        st = ctx.st
        w1, b1, w2, b2 = ctx.names
        x2, pre, a = ctx.saved_tensors
        W1, W2 = st.w(w1), st.w(w2)
        dy2 = dy.reshape(-1, W2.shape[0]).contiguous()
        dpre = _dx(st, w2, W2.shape, dy2, mulgrad=pre, act=ctx.act)
        _wgrad(st, w2, dy2, a, W2.shape)
        _bgrad(st, b2, dy2)
        _wgrad(st, w1, dpre, x2, W1.shape)
        _bgrad(st, b1, dpre)
        dx = _dx(st, w1, W1.shape, dpre).view(ctx.xshape) if ctx.needs_input_grad[0] else None
        return dx, None, None, None, None, None, None, None


class NormFn(_StoreFn):
    """stand-alone LayerNorm / RMSNorm: CLIP pre_layrnorm, Qwen2 final norm, DiT final LayerNorm."""

    @staticmethod
    def forward(ctx, x, anchor, st: ParamStore, kind: str, wn: Optional[str], bn: Optional[str], eps: float):
        if kind == "rms":
            y, rstd = K.rmsnorm_fwd(x.contiguous(), st.w(wn), eps)
            mean = rstd.new_empty(0)
        elif kind == "rms1p":                       # GemmaRMSNorm: scale by (1 + weight) in fp32
            y, rstd = K.rmsnorm_fwd(x.contiguous(), st.w(wn).float() + 1.0, eps)
            mean = rstd.new_empty(0)
        else:
            y, mean, rstd = K.layernorm_fwd(x.contiguous(), st.w(wn) if wn else None, st.w(bn) if bn else None, eps)
        ctx.st, ctx.kind, ctx.wn, ctx.bn = st, kind, wn, bn
        _use(ctx, st, wn, bn if kind not in ("rms", "rms1p") else None)
        ctx.save_for_backward(x, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        st = ctx.st
        x, mean, rstd = ctx.saved_tensors
        dy = dy.contiguous()
        if ctx.kind in ("rms", "rms1p"):
            tr = st.trainable(ctx.wn)
            w = st.w(ctx.wn) if ctx.kind == "rms" else st.w(ctx.wn).float() + 1.0      # d(1+w) = dw
            dx, _ = K.rmsnorm_bwd(dy, x.contiguous(), w, rstd, dw_out=st.g(ctx.wn) if tr else None,
                                  accumulate=st.accum_flag(ctx.wn), want_dw=tr)
            if tr:
                st.mark_written(ctx.wn)
        else:
            dx = _ln_bwd(st, dy, x.contiguous(), ctx.wn, ctx.bn, mean, rstd).view(x.shape)
        return dx, None, None, None, None, None, None


class VitEmbedFn(_StoreFn):
    """CLS + position embeddings (HF:clip/modeling_clip.py:206-217)."""

    @staticmethod
    def forward(ctx, patch, anchor, st: ParamStore, cls_n: str, pos_n: str, N: int, np_: int):
        x = K.vit_embed_fwd(patch.contiguous(), st.w(cls_n), st.w(pos_n), N, np_)
        ctx.st, ctx.cls_n, ctx.pos_n, ctx.N, ctx.np_ = st, cls_n, pos_n, N, np_
        _use(ctx, st, cls_n, pos_n)
        return x

    @staticmethod
    def backward(ctx, dx):
        st, N, np_ = ctx.st, ctx.N, ctx.np_
        dx = dx.contiguous()
        C_ = dx.shape[-1]
        dpatch = K.vit_embed_bwd(dx, N, np_)
        if st.trainable(ctx.pos_n) or st.trainable(ctx.cls_n):
            s = K.colsum(dx.view(N, (np_ + 1) * C_))
            _vgrad(st, ctx.pos_n, s)
            _vgrad(st, ctx.cls_n, s[:C_])
        return dpatch, None, None, None, None, None, None


class DropClsFn(Function):
    """feature_select: hidden_states[-2][:, 1:] (mm_vision/clip/clip_encoder.py:31-36)."""

    @staticmethod
    def forward(ctx, x):
        N, T, C_ = x.shape
        ctx.shape = (N, T, C_)
        return K.vit_embed_bwd(x.contiguous(), N, T - 1).view(N, T - 1, C_)

    @staticmethod
    def backward(ctx, dy):
        N, T, C_ = ctx.shape
        dx = torch.zeros((N, T * C_), device=dy.device, dtype=dy.dtype)
        K.copy2d(dy.reshape(N, (T - 1) * C_).contiguous(), dx[:, C_:], (T - 1) * C_, (T - 1) * C_)
        return dx.view(N, T, C_)


class SpliceFn(_StoreFn):
    """_prepare_inputs_labels_for_multimodal (dexbotic_arch.py:182-373): embed_tokens gather + image
    block insertion + zero padding, driven by the integer plan built on the host (splice.py)."""

    @staticmethod
    def forward(ctx, img_feats, anchor, st: ParamStore, embed_n: str, plan: torch.Tensor):
        d = img_feats.shape[-1]
        out = K.splice_fwd(plan, st.w(embed_n), img_feats.reshape(-1, d).contiguous())
        ctx.st, ctx.embed_n, ctx.ishape = st, embed_n, img_feats.shape
        _use(ctx, st, embed_n)
        ctx.save_for_backward(plan)
        return out

    @staticmethod
    def backward(ctx, dout):
        st = ctx.st
        (plan,) = ctx.saved_tensors
        dout = dout.contiguous()
        d_img = None
        if ctx.needs_input_grad[0]:
            d_img = torch.zeros(ctx.ishape, device=dout.device, dtype=dout.dtype)
        d_embed = None
        if st.trainable(ctx.embed_n):
            d_embed = st.g(ctx.embed_n)
            if not st.accum_flag(ctx.embed_n):
                # dense nn.Embedding gradient of which only <= B*S_text rows are ever non-zero: re-zero just the rows
                # the previous step touched when the store tracks them (single GPU), the whole slice otherwise
                st.zero_embed_grad(ctx.embed_n)
            st.note_embed_rows(ctx.embed_n, plan)
        K.splice_bwd(plan, dout, d_embed, d_img)
        if d_embed is not None:
            # AFTER the scatter-add is enqueued: embed_tokens is alone in its bucket, so this fires the reducer / the
            # grad-norm fold, which order themselves after everything enqueued so far on this stream
            st.mark_written(ctx.embed_n)
        return d_img, None, None, None, None


class GatherRowsFn(Function):
    """cognition feature = hidden state of the last un-padded token (cogact_arch.py:110-120); output fp32
    (the action head runs in fp32, cogact_arch.py:133)."""

    @staticmethod
    def forward(ctx, x, idx):
        ctx.R, ctx.dtype = x.shape[0], x.dtype
        ctx.save_for_backward(idx)
        return K.gather_rows(x.contiguous(), idx, torch.float32)

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        return K.scatter_rows(dout.contiguous(), idx, ctx.R, ctx.dtype), None


class TokenDropFn(_StoreFn):
    """LabelEmbedder.token_drop (dit.py:80-96)."""

    @staticmethod
    def forward(ctx, z, anchor, st: ParamStore, unc_n: str, drop: torch.Tensor):
        ctx.st, ctx.unc_n = st, unc_n
        _use(ctx, st, unc_n)
        ctx.save_for_backward(drop)
        return K.token_drop(z.contiguous(), st.w32(unc_n).view(-1), drop)

    @staticmethod
    def backward(ctx, dout):
        (drop,) = ctx.saved_tensors
        dz, dunc = K.token_drop_bwd(dout.contiguous(), drop, want_dz=ctx.needs_input_grad[0])
        _vgrad(ctx.st, ctx.unc_n, dunc)
        return dz, None, None, None, None


class DitAssembleFn(_StoreFn):
    """x = cat(t_emb + z_emb, x_emb) + positional_embedding (dit.py:281-286)."""

    @staticmethod
    def forward(ctx, xe, te, ze, anchor, st: ParamStore, pos_n: str):
        ctx.st, ctx.pos_n = st, pos_n
        _use(ctx, st, pos_n)
        return K.dit_assemble_fwd(xe.contiguous(), te.contiguous(), ze.contiguous(), st.w32(pos_n))

    @staticmethod
    def backward(ctx, dh):
        dh = dh.contiguous()
        N, T1, hd = dh.shape
        dxe, dc = K.dit_assemble_bwd(dh)
        if ctx.st.trainable(ctx.pos_n):
            _vgrad(ctx.st, ctx.pos_n, K.colsum(dh.view(N, T1 * hd)))
        return dxe, dc, dc, None, None, None


class MseLossFn(Function):
    """loss = ((eps_hat - eps)**2).mean()  (action_models.py:119-121)."""

    @staticmethod
    def forward(ctx, pred, target):
        loss, dpred = K.mse_loss(pred.contiguous(), target.contiguous(), 1.0, want_grad=True)
        ctx.save_for_backward(dpred)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        (dpred,) = ctx.saved_tensors
        # g is the scalar upstream gradient (1.0 for loss.backward(); 1/accum under gradient
        # accumulation): applied on the device, no host sync
        return K.scale_dev_(dpred.clone(), g.reshape(1).float().contiguous()), None


LMHEAD_SLAB_BYTES = (1 << 31) - 4096        # largest output one launch of the MFMA fast path addresses (tests shrink it)


class LmHeadLossFn(_StoreFn):
    """logits = lm_head(hidden) and the HF causal-LM cross-entropy over the (already shifted) labels
    (dexbotic_arch.py:483-488; transformers/loss/loss_utils.py ForCausalLMLoss): loss = mean over the non-ignored
    rows of logsumexp(logits) - logits[label], in fp32 on the stored logits.  Returns (loss, logits).  The vocabulary
    GEMMs are the only other large contractions of the path (4592 x 152064 x 3584 at the CogACT batch): bf16 runs them
    on the ping-pong MFMA kernel — logits NT, dW = dZ^T H as a TN and dH = dZ W as an NN product, operands as they lie."""

    @staticmethod
    def forward(ctx, hidden, anchor, st: ParamStore, wn: str, labels_shifted: torch.Tensor, n_valid: int):
        h2 = hidden.reshape(-1, hidden.shape[-1]).contiguous()
        W = st.w(wn)
        logits = K.mm_nt(h2, W)
        row_loss, lse = K.cross_entropy_fwd(logits, labels_shifted)
        loss = K.colsum(row_loss.view(-1, 1))
        if n_valid > 0:
            K.scale_(loss, 1.0 / n_valid)
        else:
            loss = loss * float("nan")                          # F.cross_entropy(mean) over zero targets
        ctx.st, ctx.wn, ctx.n_valid, ctx.hshape = st, wn, n_valid, hidden.shape
        _use(ctx, st, wn)
        ctx.save_for_backward(h2, logits, lse, labels_shifted)
        ctx.mark_non_differentiable(logits)
        return loss.view(()), logits.view(*hidden.shape[:-1], W.shape[0])

    @staticmethod
    def backward(ctx, g, _g_logits):
        st, wn = ctx.st, ctx.wn
        h2, logits, lse, labels = ctx.saved_tensors
        W = st.w(wn)
        dz = K.cross_entropy_bwd(logits, labels, lse, g.reshape(1).float().contiguous(), 1.0 / max(ctx.n_valid, 1))
        if st.trainable(wn):
            # the fp32 dW of the full vocabulary (152064 x 3584 x 4 B = 2.18 GB) is past the 2 GiB the MFMA fast path addresses
            # in one output: written in row slabs of < 2 GiB (column slices of dZ), each on the fast path
            g, acc, mir = st.g(wn), st.accum_flag(wn), st.mirror_out(wn)
            V, d_ = g.shape
            slab = max(256, min(V, LMHEAD_SLAB_BYTES // (4 * d_) // 256 * 256))
            for lo in range(0, V, slab):
                hi = min(V, lo + slab)
                K.mm_tn(dz[:, lo:hi], h2, out=g[lo:hi], accumulate=acc, mirror=None if mir is None else mir[lo:hi])
            st.mark_written(wn)
        dh = None
        if ctx.needs_input_grad[0]:
            dh = K.mm_nn(dz, W).view(ctx.hshape)
        return dh, None, None, None, None, None


class AddPosFn(_StoreFn):
    """x[N,T,C] + pos[T,C] (learned position embedding of the SigLIP tower, HF siglip/modeling_siglip.py
    SiglipVisionEmbeddings); d_pos = sum over N of dy."""

    @staticmethod
    def forward(ctx, x, anchor, st: ParamStore, pos_name: str):
        N, T, C_ = x.shape
        pos = st.w(pos_name).unsqueeze(0).expand(N, T, C_).contiguous()
        ctx.st, ctx.pos_name, ctx.shape = st, pos_name, (N, T, C_)
        _use(ctx, st, pos_name)
        return K.add(x.contiguous(), pos)

    @staticmethod
    def backward(ctx, dy):
        st, (N, T, C_) = ctx.st, ctx.shape
        dy = dy.contiguous()
        if st.trainable(ctx.pos_name):
            # rows of [N, T*C]: the column sums are the per-position gradient
            K.colsum(dy.view(N, T * C_), out=st.g(ctx.pos_name).view(-1), accumulate=st.accum_flag(ctx.pos_name))
            st.mark_written(ctx.pos_name)
        return dy, None, None, None


class MseLossRowsFn(Function):
    """sample-weighted eps-MSE (hybrid_cogact_arch.py:168-173): sum_r w_r mean(d_r^2) / (sum_r w_r + 1e-6)"""

    @staticmethod
    def forward(ctx, pred, target, row_w):
        loss, dpred = K.mse_loss_rows(pred.contiguous(), target.contiguous(), row_w.float().contiguous(), 1.0, want_grad=True)
        ctx.save_for_backward(dpred)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        (dpred,) = ctx.saved_tensors
        return K.scale_dev_(dpred.clone(), g.reshape(1).float().contiguous()), None, None


class ScaleFn(Function):
    """y = x * s in the tensor's dtype (token embeddings * sqrt(hidden), pi0_arch.py:247-250)"""

    @staticmethod
    def forward(ctx, x, s: float):
        ctx.s = s
        return ScaleFn._mul(x, s)

    @staticmethod
    def _mul(x, s):
        x = x.contiguous()
        if x.dtype == torch.float32:
            return K.scale_(x.clone(), s)
        return K.cast(K.scale_(K.cast(x, torch.float32), s), x.dtype)

    @staticmethod
    def backward(ctx, dy):
        return ScaleFn._mul(dy, ctx.s), None


@dataclass
class GemmaLayerSpec:
    """parameter names + shapes of one Gemma decoder layer of one expert (model/llm/gemma.py)"""
    ln1: str
    qkv: Tuple[str, ...]
    o: str
    ln2: str
    gu: Tuple[str, ...]
    down: str
    d: int
    F: int
    eps: float


def _gemma_norm_w(st: ParamStore, name: str) -> torch.Tensor:
    """fp32 (1 + w) of a GemmaRMSNorm weight, computed once per state of the weights (forward and backward of a step share it; the
    two aten launches per use were 290 launches of a pi0 step)"""
    key = st.weights_key()
    cache = st.__dict__.setdefault("_gemma_norm_cache", {})
    if cache.get("key") != key:
        cache.clear()
        cache["key"] = key
    w = cache.get(name)
    if w is None:
        w = cache[name] = st.w(name).float() + 1.0
    return w


class Pi0MotLayerFn(_StoreFn):
    """One layer of the pi0 mixture of transformers (pi0_arch.py:130-216) for its two experts at once: per expert
    GemmaRMSNorm -> fused q/k/v -> RoPE; ONE attention over the concatenated tokens with the block-prefix mask; per
    expert o_proj + residual -> GemmaRMSNorm -> GeGLU -> residual.  Inputs/outputs are the experts' [B*S_e, d_e]
    activations.  ``skip_post0``: the last layer's llm half after attention feeds only prefix_out, which the loss
    never reads — it is not computed (the reference computes it and its parameters get no gradient)."""

    @staticmethod
    def _run(st: ParamStore, sps, geom, skip_post0: bool, x0, x1, cos_t, sin_t, pos0, pos1, q_limit, key_valid):
        """the layer's forward launches -> ((y0, y1), what the backward reads besides x0, x1)"""
        B, S0, S1, Hq, Hkv, D = geom
        nq = (Hq + 2 * Hkv) * D
        xs, Ss, poss = (x0, x1), (S0, S1), (pos0, pos1)
        h1, rstd1 = [], []
        # both experts' RoPE launches write straight into ONE q / k / v (round 6: no concatenation around the shared attention call)
        q = torch.empty((B, Hq, S0 + S1, D), device=x0.device, dtype=x0.dtype)
        k = torch.empty((B, Hkv, S0 + S1, D), device=x0.device, dtype=x0.dtype)
        v = torch.empty((B, Hkv, S0 + S1, D), device=x0.device, dtype=x0.dtype)
        off = 0
        for x, sp, S, pos in zip(xs, sps, Ss, poss):
            h, r = K.rmsnorm_fwd(x, _gemma_norm_w(st, sp.ln1), sp.eps)
            K.rope_split_into(K.mm_nt(h, st.w(*sp.qkv, shape=(nq, sp.d))), q, k, v, off, cos_t, sin_t, pos, B, S, Hq, Hkv, D)
            off += S
            h1.append(h); rstd1.append(r)
        o = torch.empty((B, S0 + S1, Hq, D), device=x0.device, dtype=x0.dtype)
        lse = K.attn_fwd(q, k, v, o.permute(0, 2, 1, 3), causal=False, scale=D ** -0.5, q_limit=q_limit, key_valid=key_valid)
        ys, saved = [], []
        off = 0
        for i, (x, sp, S) in enumerate(zip(xs, sps, Ss)):
            a = o[:, off:off + S].reshape(B * S, Hq * D).contiguous()
            off += S
            if i == 0 and skip_post0:
                ys.append(x.new_zeros((0,)))                   # placeholder, never read downstream
                saved += [a, a.new_empty(0), a.new_empty(0), a.new_empty(0), a.new_empty(0), a.new_empty(0)]
                continue
            r = K.mm_nt(a, st.w(sp.o), residual=x)
            h2, rs2 = K.rmsnorm_fwd(r, _gemma_norm_w(st, sp.ln2), sp.eps)
            gu = K.mm_nt(h2, st.w(*sp.gu, shape=(2 * sp.F, sp.d)))
            act = K.glu_fwd(gu, L.ACT_GELU_TANH)
            ys.append(K.mm_nt(act, st.w(sp.down), residual=r))
            saved += [a, r, rs2, h2, gu, act]
        return (ys[0], ys[1]), (h1[0], h1[1], rstd1[0], rstd1[1], q, k, v, o, lse, *saved)

    @staticmethod
    def forward(ctx, x0, x1, anchor, st: ParamStore, sp0: GemmaLayerSpec, sp1: GemmaLayerSpec, geom, cos_t, sin_t,
                pos0, pos1, q_limit, key_valid, skip_post0: bool):
        sps = (sp0, sp1)
        ys, saved = Pi0MotLayerFn._run(st, sps, geom, skip_post0, x0, x1, cos_t, sin_t, pos0, pos1, q_limit, key_valid)
        ctx.st, ctx.sps, ctx.geom, ctx.skip_post0 = st, sps, geom, skip_post0
        for i, sp in enumerate(sps):
            _use(ctx, st, sp.ln1, sp.qkv)
            if not (i == 0 and skip_post0):
                _use(ctx, st, sp.o, sp.ln2, sp.gu, sp.down)
        ctx.aux = (cos_t, sin_t, pos0, pos1, q_limit, key_valid)
        ctx.recompute = _recompute(ctx, st)
        if ctx.recompute:
            ctx.save_for_backward(x0, x1)
        else:
            ctx.save_for_backward(x0, x1, *saved)
        return ys

    @staticmethod
    def backward(ctx, dy0, dy1):
        st, sps, skip_post0 = ctx.st, ctx.sps, ctx.skip_post0
        B, S0, S1, Hq, Hkv, D = ctx.geom
        cos_t, sin_t, pos0, pos1, q_limit, key_valid = ctx.aux
        sv = ctx.saved_tensors
        if ctx.recompute:
            sv = tuple(sv) + tuple(Pi0MotLayerFn._run(st, sps, ctx.geom, skip_post0, sv[0], sv[1], cos_t, sin_t, pos0, pos1,
                                                       q_limit, key_valid)[1])
        xs, h1, rstd1 = sv[0:2], sv[2:4], sv[4:6]
        q, k, v, o, lse = sv[6:11]
        per = [sv[11:17], sv[17:23]]
        nq = (Hq + 2 * Hkv) * D
        Ss, poss, dys = (S0, S1), (pos0, pos1), (dy0, dy1)
        S = S0 + S1
        do = torch.zeros((B, S, Hq, D), device=q.device, dtype=q.dtype)
        drs = []
        off = 0
        for i, (sp, Sn, dy) in enumerate(zip(sps, Ss, dys)):
            a, r, rs2, h2, gu, act = per[i]
            if i == 0 and skip_post0:
                drs.append(None)
                off += Sn
                continue
            dy = dy.contiguous()
            dact = _dx(st, sp.down, (sp.d, sp.F), dy)
            _wgrad(st, sp.down, dy, act, (sp.d, sp.F))
            dgu = K.glu_bwd(gu, dact, L.ACT_GELU_TANH)
            dh2 = _dx(st, sp.gu, (2 * sp.F, sp.d), dgu)
            _wgrad(st, sp.gu, dgu, h2, (2 * sp.F, sp.d))
            tr = st.trainable(sp.ln2)
            dr, _ = K.rmsnorm_bwd(dh2, r, _gemma_norm_w(st, sp.ln2), rs2, dw_out=st.g(sp.ln2) if tr else None,
                                  accumulate=st.accum_flag(sp.ln2), want_dw=tr, residual=dy)
            if tr:
                st.mark_written(sp.ln2)
            da = _dx(st, sp.o, (sp.d, Hq * D), dr)
            _wgrad(st, sp.o, dr, a, (sp.d, Hq * D))
            do[:, off:off + Sn].copy_(da.view(B, Sn, Hq, D))
            off += Sn
            drs.append(dr)
        do_h = K.permute_bshd(do, B, S, Hq, D, True)                       # head-major dO for the GQA fold
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        K.attn_bwd(q, k, v, o.permute(0, 2, 1, 3), lse, do_h, dq, dk, dv, causal=False, scale=D ** -0.5,
                   q_limit=q_limit, key_valid=key_valid)
        dxs = []
        off = 0
        for i, (sp, Sn, pos) in enumerate(zip(sps, Ss, poss)):
            dqkv = K.rope_merge_from(dq, dk, dv, off, cos_t, sin_t, pos, B, Sn, Hq, Hkv, D)      # (no slicing copies)
            off += Sn
            dh = _dx(st, sp.qkv, (nq, sp.d), dqkv)
            _wgrad(st, sp.qkv, dqkv, h1[i], (nq, sp.d))
            tr = st.trainable(sp.ln1)
            dxn, _ = K.rmsnorm_bwd(dh, xs[i], _gemma_norm_w(st, sp.ln1), rstd1[i], dw_out=st.g(sp.ln1) if tr else None,
                                   accumulate=st.accum_flag(sp.ln1), want_dw=tr, residual=drs[i])
            if tr:
                st.mark_written(sp.ln1)
            dxs.append(dxn)
        return (dxs[0], dxs[1]) + (None,) * 12


# ------------------------------------------------------------------ small differentiable pieces (MemVLA memory modules)
class AddFn(Function):
    """a + b (same shape): residual connections outside the fused blocks"""

    @staticmethod
    def forward(ctx, a, b):
        return K.add(a.contiguous(), b.contiguous())

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


class ForkFn(Function):
    """x -> n aliases of x, one per consumer: the sum of the consumers' gradients is then library launches (dxa_add) instead of
    the autograd engine's own at::add on a tensor that fans out (the MemVLA retrieval path is built from small Functions)"""

    @staticmethod
    def forward(ctx, x, n=2):
        ctx.n = n
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        gs = [g for g in gs if g is not None]
        if not gs:
            return None, None
        acc = gs[0].contiguous()
        for g in gs[1:]:
            acc = K.add(acc, g.contiguous())
        return acc, None


class CatLastFn(Function):
    """[R, Da] , [R, Db] -> [R, Da + Db] (torch.cat(dim=-1) of the gate-fusion input, memvla_arch.py:176-180) with the
    library's 2-D copy; the backward hands the two column blocks back as contiguous tensors"""

    @staticmethod
    def forward(ctx, a, b):
        R, Da = a.shape
        Db = b.shape[1]
        ctx.dims = (Da, Db)
        out = torch.empty((R, Da + Db), device=a.device, dtype=a.dtype)
        K.copy2d(a.contiguous(), out[:, :Da], Da, Da)
        K.copy2d(b.contiguous(), out[:, Da:], Db, Db)
        return out

    @staticmethod
    def backward(ctx, dy):
        Da, Db = ctx.dims
        R = dy.shape[0]
        da = torch.empty((R, Da), device=dy.device, dtype=dy.dtype)
        db = torch.empty((R, Db), device=dy.device, dtype=dy.dtype)
        K.copy2d(dy[:, :Da], da, Da, Da)
        K.copy2d(dy[:, Da:], db, Db, Db)
        return da, db


class AttnFn(Function):
    """softmax(q k^T / sqrt(D)) v for [B,S,H,D] VIEWS (any strides with D contiguous) -> [B,Sq,H*D]: the cross
    attention of CrossTransformerBlock (memvla_arch.py:84-127) and of the DiT per-attention (nn.MultiheadAttention,
    memvla/action_model/dit.py:158-185).  ``drop_mask`` ([B,H,Sq,Sk], entries 0 or 1/(1-p)): SDPA's dropout_p on the
    attention weights (memvla_arch.py:120-123), the mask drawn by the caller."""

    @staticmethod
    def forward(ctx, q, k, v, drop_mask=None):
        B, Sq, H, D = q.shape
        o = torch.empty((B, Sq, H, D), device=q.device, dtype=q.dtype)
        lse = K.attn_fwd(q.permute(0, 2, 1, 3), k.permute(0, 2, 1, 3), v.permute(0, 2, 1, 3), o.permute(0, 2, 1, 3),
                         causal=False, scale=D ** -0.5, drop_mask=drop_mask)
        ctx.save_for_backward(q, k, v, o, lse)
        ctx.drop_mask = drop_mask
        return o.view(B, Sq, H * D)

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse = ctx.saved_tensors
        B, Sq, H, D = q.shape
        do = do.contiguous().view(B, Sq, H, D)
        dq, dk, dv = torch.empty(q.shape, device=q.device, dtype=q.dtype), torch.empty(k.shape, device=q.device, dtype=q.dtype), \
            torch.empty(v.shape, device=q.device, dtype=q.dtype)
        P = lambda t: t.permute(0, 2, 1, 3)
        K.attn_bwd(P(q), P(k), P(v), P(o), lse, P(do), P(dq), P(dk), P(dv), causal=False, scale=D ** -0.5,
                   drop_mask=ctx.drop_mask)
        return dq, dk, dv, None


class AttnPackedFn(Function):
    """AttnFn on PACKED projections: ``qkv`` [B,S,3,H,D] (self attention, q = None) or ``q`` [B,Sq,H,D] + ``kv`` [B,Sk,2,H,D]
    (nn.MultiheadAttention's packed in_proj of the DiT's perceptual cross attention, memvla/action_model/dit.py:158-185).  The
    backward writes dq / dk / dv straight into ONE packed gradient buffer (the attention kernels take strided views), so
    autograd sees a single tensor with a single consumer: no select_backward zero-fills, no gradient adds."""

    @staticmethod
    def forward(ctx, q, packed):
        self_attn = q is None
        ctx.self_attn = self_attn
        if self_attn:
            q, k, v = packed[:, :, 0], packed[:, :, 1], packed[:, :, 2]
        else:
            k, v = packed[:, :, 0], packed[:, :, 1]
        B, Sq, H, D = q.shape
        o = torch.empty((B, Sq, H, D), device=q.device, dtype=q.dtype)
        P = lambda t: t.permute(0, 2, 1, 3)
        lse = K.attn_fwd(P(q), P(k), P(v), P(o), causal=False, scale=D ** -0.5)
        ctx.save_for_backward(q if not self_attn else packed, packed, o, lse)
        return o.view(B, Sq, H * D)

    @staticmethod
    def backward(ctx, do):
        qs, packed, o, lse = ctx.saved_tensors
        dpk = torch.empty(packed.shape, device=packed.device, dtype=packed.dtype)
        if ctx.self_attn:
            q, k, v = packed[:, :, 0], packed[:, :, 1], packed[:, :, 2]
            dq, dk, dv = dpk[:, :, 0], dpk[:, :, 1], dpk[:, :, 2]
        else:
            q, k, v = qs, packed[:, :, 0], packed[:, :, 1]
            dq, dk, dv = torch.empty(q.shape, device=q.device, dtype=q.dtype), dpk[:, :, 0], dpk[:, :, 1]
        B, Sq, H, D = q.shape
        do = do.contiguous().view(B, Sq, H, D)
        P = lambda t: t.permute(0, 2, 1, 3)
        K.attn_bwd(P(q), P(k), P(v), P(o), lse, P(do), P(dq), P(dk), P(dv), causal=False, scale=D ** -0.5)
        return (None, dpk) if ctx.self_attn else (dq, dpk)


class DropFn(Function):
    """x * mask with mask entries 0 or 1/(1-p): nn.Dropout with the mask drawn by the caller (the FFN of the retrieval
    blocks, memvla_arch.py:99-105)"""

    @staticmethod
    def forward(ctx, x, mask):
        ctx.save_for_backward(mask)
        return K.mul(x.contiguous(), mask)

    @staticmethod
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        return K.mul(dy.contiguous(), mask), None


class GateFuseFn(Function):
    """scale * x1 + (1 - scale) * x2 (GateFusion, memvla_arch.py:170-187; scale = sigmoid(proj(cat)) comes in)"""

    @staticmethod
    def forward(ctx, scale, x1, x2):
        scale, x1, x2 = scale.contiguous(), x1.contiguous(), x2.contiguous()
        d = K.axpby(x1, x2, 1.0, -1.0)
        ctx.save_for_backward(scale, d)
        return K.add(K.mul(scale, d), x2)

    @staticmethod
    def backward(ctx, dy):
        scale, d = ctx.saved_tensors
        dy = dy.contiguous()
        dd = K.mul(dy, scale)
        return K.mul(dy, d), dd, K.axpby(dy, dd, 1.0, -1.0)


class TokenMeanFn(Function):
    """mean over the token axis: [B,N,C] -> [B,C] (AdaptiveAvgPool2d(1) of BottleneckSE, memvla_arch.py:139-141)"""

    @staticmethod
    def forward(ctx, x):
        B, N, C_ = x.shape
        ctx.shape = (B, N, C_)
        return K.token_sum(x.contiguous(), 1.0 / N)          # one launch: fp32 sums in token order, scaled, rounded once

    @staticmethod
    def backward(ctx, dm):
        B, N, C_ = ctx.shape
        return K.add_rows(None, dm.contiguous(), N, 1.0 / N)  # dm / N broadcast over the tokens


class AddRowsFn(Function):
    """x [R, N, C] + g [R, C] broadcast over the tokens: the timestep positional embedding on every token of a memory entry
    (memvla_arch.py:352-360: pe.unsqueeze(1).expand(-1, N, -1) added to the bank)"""

    @staticmethod
    def forward(ctx, x, g):
        ctx.n = x.shape[1]
        ctx.needs = (x.requires_grad, g.requires_grad)
        return K.add_rows(x.contiguous(), g.contiguous(), x.shape[1])

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        return (dy if ctx.needs[0] else None), (K.token_sum(dy) if ctx.needs[1] else None)


class UnbindRowsFn(Function):
    """[B, ...] -> B tensors [1, ...] (the per-sample walk of the memory bank); the backward writes the B gradients side by
    side with library copies instead of autograd's zero-fill + add per slice"""

    @staticmethod
    def forward(ctx, x):
        ctx.shape, ctx.dt = tuple(x.shape), x.dtype
        x = x.contiguous()
        return tuple(x[i:i + 1] for i in range(x.shape[0]))

    @staticmethod
    def backward(ctx, *gs):
        out = torch.empty(ctx.shape, device=next(g for g in gs if g is not None).device, dtype=ctx.dt)
        for i, g in enumerate(gs):
            if g is None:
                out[i].zero_()                                   # (a sample whose output nobody used: not on the training path)
            else:
                K.cast(g.contiguous(), ctx.dt, out=out[i:i + 1])
        return out


class CatRowsFn(Function):
    """B tensors [1, ...] -> [B, ...] with library copies (torch.cat of the per-sample outputs)"""

    @staticmethod
    def forward(ctx, *xs):
        out = torch.empty((len(xs),) + tuple(xs[0].shape[1:]), device=xs[0].device, dtype=xs[0].dtype)
        for i, x in enumerate(xs):
            K.cast(x.contiguous(), out.dtype, out=out[i:i + 1])
        return out

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        return tuple(dy[i:i + 1] for i in range(dy.shape[0]))


class RowGateFn(Function):
    """x[B,N,C] * g[B,C] (SE channel gate, memvla_arch.py:160-161)"""

    @staticmethod
    def forward(ctx, x, g):
        x, g = x.contiguous(), g.contiguous()
        ctx.save_for_backward(x, g)
        return K.mul_rows(x, g)

    @staticmethod
    def backward(ctx, dy):
        x, g = ctx.saved_tensors
        dy = dy.contiguous()
        B = x.shape[0]
        return K.mul_rows(dy, g), K.token_sum(K.mul(dy, x))
