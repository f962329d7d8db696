"""Flat parameter arenas, fused optimizer step and the data-parallel gradient reducer.

MI355X-first memory plan (288 GB HBM3E per GPU): every trainable tensor of the policy lives in ONE
fp32 master arena; a bf16 shadow arena (what the MFMA kernels read), an fp32 gradient arena and the
two Adam moment arenas have the same layout.  ``nn.Parameter``s are views into the master arena, so
``state_dict()`` / ``load_state_dict()`` keep the reference's key map while
  * the optimizer is ONE fused multi-tensor AdamW launch over the arena (dxa_adamw) that also
    refreshes the bf16 shadows,
  * the global grad-norm is one reduction over the gradient arena,
  * data-parallel gradient averaging all-reduces contiguous slices of the gradient arena in place
    (no flatten/unflatten copies) on a side HIP stream, bucket by bucket, as soon as the backward of
    the owning block has written them.
Adjacent members of a group (q/k/v projections, gate/up) are packed back to back so the fused
[q;k;v] / [gate;up] matrices are plain views too.

Reference counterparts: OptimizerConfig._get_optimizer_grouped_parameters (exp/base_exp.py:95-203),
DexboticTrainer.create_optimizer / _link_exp_config (exp/trainer.py:25-36,88-124: AdamW, betas
(0.9,0.999), eps 1e-8, max_grad_norm 1.0), DDP / ZeRO gradient sync (exp/trainer.py:110,121).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Callable, Dict, Iterable, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

ALIGN = 64  # elements; every group starts on a 256-byte boundary

# The reference's factories take ONE argument (build_vision_tower(cfg), build_vision_projector(config),
# build_action_model(config): mm_vision/builder.py:9-34, mm_projector/builder.py:36-81, cogact/action_model/builder.py:5-27).
# Native modules additionally need the arena they register their parameters in: the *ForCausalLM constructor opens a
# build context and the factories pick the store up from it.
_BUILD_STACK: List["ParamStore"] = []


class building:
    """``with building(store): ...`` — every factory called inside registers into ``store``"""

    def __init__(self, store: "ParamStore"):
        self.store = store

    def __enter__(self):
        _BUILD_STACK.append(self.store)
        return self.store

    def __exit__(self, *exc):
        _BUILD_STACK.pop()
        return False


def current_store(store: Optional["ParamStore"] = None) -> "ParamStore":
    if store is not None:
        return store
    if not _BUILD_STACK:
        raise RuntimeError("no ParamStore: native modules are built inside a *ForCausalLM constructor "
                           "(or `with dexbotic_amd.engine.building(store):`)")
    return _BUILD_STACK[-1]


@dataclass
class Slot:
    name: str
    shape: Tuple[int, ...]
    offset: int
    numel: int
    bucket: int = 0


_HOOK_JOINS = __import__('os').environ.get('DXA_JOIN_AT_BUCKET', '0') != '1'     # 1: rounds-4 behaviour (A/B)


class ParamStore:
    """Owns the arenas.  ``specs``: ordered list of groups; a group is a list of (name, shape) packed
    contiguously; ``bucket`` ids follow registration order (= forward order) and are the unit of the
    DP all-reduce."""

    def __init__(self, device: torch.device | str, compute_dtype: torch.dtype = torch.float32):
        self.device = torch.device(device)
        self.compute_dtype = compute_dtype
        self.slots: Dict[str, Slot] = {}
        self.epi_sumsq = False                  # the trainer turns it on: one GPU, clip enabled
        # activation recompute (the reference's gradient checkpointing, base_exp.py:245): the transformer-layer Functions keep
        # only their input and re-run their forward inside their backward (functional._recompute).  Off: activations resident.
        self.recompute = False
        # weight-gradient products on a side HIP stream (functional._wgrad_now): dW = dY^T X depends on nothing downstream in the
        # backward chain, so it may run beside the next dX product; the compute stream joins (join_wgrad) before anything reads the
        # gradient arena — a bucket's completion hook, the norm / reducer finish, the optimizer.  None: same stream (default).
        self.wgrad_stream: Optional["torch.cuda.Stream"] = None
        self.wgrad_stream_f32_only = False      # only the fp32 action head's dW products (low occupancy: 1088 rows) go there
        self.bgrad_on_side = False              # the bias gradients' column sums too
        self._wgrad_pending = False
        # gradient accumulation: only the LAST micro-batch's dW products (accumulate = 1: what they store is the step's final
        # gradient) leave their share of sum(g^2); the trainer keeps this flag current (True without accumulation)
        self.last_micro = True
        # two micro-batches per optimizer step: the weight gradients of single-use linears are ONE product over both micro-
        # batches' (dY, X) pairs (functional._wgrad; dxa_gemm_desc.A2 / B2) instead of a read-modify-write of the fp32 gradient
        self.accum_merge = False
        self._accum_stash: Dict[tuple, tuple] = {}
        # weight gradients of a parameter applied k times in one forward (MemVLA's per-sample retrieval blocks): the k
        # (dY, X) pairs are collected and ONE product over all their rows writes dW once (functional._wgrad), instead of k
        # read-modify-write passes over the gradient (k rank-1 updates of a 14336 x 3584 matrix for the one-token cognition
        # stream).  The trainer turns it on and flushes what a pruned backward left behind.
        self.defer_wgrad = False
        self._wg_stash: Dict[tuple, dict] = {}
        self._bg_stash: Dict[tuple, dict] = {}     # the same for the bias gradients: one column sum over all consumers' dY (functional._bgrad)
        # bf16 gradient arena (the reference's DeepSpeed bf16 recipe, script/deepspeed/zero3.json "bf16": gradients are bf16,
        # the optimizer keeps fp32 masters): the bf16 dW products write ONLY the bf16 arena (gradc) — half the epilogue bytes and
        # half of AdamW's gradient read.  Gradients other kernels produce (norm weights, biases, embeddings, the fp32 head) land
        # in the fp32 arena and are cast per slot when their bucket completes (GradReducer._fill_mirror).
        self.bf16_grads = False
        self._ssq_buf: Optional[torch.Tensor] = None
        self._ssq_cursor = 0
        self._ssq_covered: set = set()
        self._excluded: set = set()
        self._slot_order = None
        self._slot_starts = None
        self.layernorm_modules: set = set()          # dotted module paths that are nn.LayerNorm in the reference
        self._groups: List[List[str]] = []
        self._cursor = 0
        self._bucket = 0
        self.bucket_ranges: List[List[int]] = []   # per bucket [lo, hi)
        self.master: Optional[torch.Tensor] = None
        self.shadow: Optional[torch.Tensor] = None
        self.grad: Optional[torch.Tensor] = None
        self.gradc: Optional[torch.Tensor] = None     # bf16 communication copy of the gradient arena (DP, bf16 exchange)
        self._mirrored: set = set()                   # slots whose copy a kernel epilogue already wrote this micro-batch
        self.params: Dict[str, nn.Parameter] = {}
        self.grad_written: Dict[str, bool] = {}
        self.on_bucket_ready: Optional[Callable[[int], None]] = None
        self._bucket_pending: List[int] = []
        self._micro_written: set = set()
        self._micro_touched: set = set()
        self._uses: Dict[str, int] = {}
        # embedding-gradient rows written since the slice was last all-zero (sparse re-zero, functional.SpliceFn)
        self.sparse_embed_zero = False
        self._embed_dirty: Dict[str, Optional[List[torch.Tensor]]] = {}     # missing / None = unknown -> dense zero
        # optimizer overlapped with the next forward (FusedAdamW(overlap=True)): bucket -> HIP event recorded on the optimizer
        # stream after that bucket's parameters / shadows / moments were updated.  Any arena view of the bucket handed out
        # afterwards (w / w32 / g / gc) first makes the CURRENT stream wait for the event, so layer i of the next step starts
        # as soon as bucket i is done while the HBM-bound update of the later buckets still runs under the MFMA-bound forward.
        self._pending: Dict[int, "torch.cuda.Event"] = {}
        self._w32_buckets: set = set()             # buckets some TRAINING forward read through w32() (the fp32 action head)
        self._record_w32 = False                   # up while a trainer's forward + backward runs (initialisers / loaders use w32() too)
        # ---- driven from outside (HF Trainer / the reference's DexboticTrainer / a hand-written loop) -----------------------
        # managed: a NativeTrainer (or exp.trainer.NativeDexboticTrainer) calls begin_step / begin_micro itself.  Otherwise the
        # model's forward pre-hook does (external_prelude): gradients re-attached after zero_grad(set_to_none=True), a new
        # optimizer step recognised by the dropped / touched gradients, and the bf16 shadows re-derived when a foreign
        # optimizer (torch.optim.AdamW on the arena views) moved the fp32 masters (w(): version counter of the arena).
        self.managed = False
        self._shadow_version: Optional[int] = None
        self._view_cache: Dict[tuple, torch.Tensor] = {}
        self.native_epoch = 0                      # bumped by every native update of the masters (FusedAdamW): with master._version the
                                                   # key of caches derived from the weights (weights_key())
        self._grad_version: Optional[int] = None
        self._external_fresh = True                     # nothing written since the last begin_step
        self._external_trained = False                  # a training forward ran under an external loop since the last sync

    # ---- layout -------------------------------------------------------------------------------------
    def new_bucket(self) -> int:
        self._bucket += 1
        return self._bucket

    def register(self, group: Sequence[Tuple[str, Sequence[int]]], layernorm: bool = False) -> None:
        """``layernorm``: the group is the weight / bias of an ``nn.LayerNorm`` of the reference: its module container
        becomes an ``nn.LayerNorm`` instance, so code that selects parameters by module TYPE (the no-weight-decay rule of
        OptimizerConfig._get_optimizer_grouped_parameters, base_exp.py:101-102: get_parameter_names(model,
        ALL_LAYERNORM_LAYERS)) treats the native model like the reference's."""
        assert self.master is None, "register() after finalize()"
        if layernorm:
            for name, _ in group:
                self.layernorm_modules.add(name.rsplit(".", 1)[0])
        self._cursor = (self._cursor + ALIGN - 1) // ALIGN * ALIGN
        names = []
        for name, shape in group:
            assert name not in self.slots, name
            n = int(math.prod(shape))
            self.slots[name] = Slot(name, tuple(int(s) for s in shape), self._cursor, n, self._bucket)
            self._cursor += n
            names.append(name)
        self._groups.append(names)

    def finalize(self, train: bool = True) -> None:
        total = (self._cursor + ALIGN - 1) // ALIGN * ALIGN
        self.total = total
        self.master = torch.zeros(total, device=self.device, dtype=torch.float32)
        if self.compute_dtype != torch.float32:
            self.shadow = torch.zeros(total, device=self.device, dtype=self.compute_dtype)
        if train:
            self.grad = torch.zeros(total, device=self.device, dtype=torch.float32)
        nb = self._bucket + 1
        self.bucket_ranges = [[total, 0] for _ in range(nb)]
        for s in self.slots.values():
            r = self.bucket_ranges[s.bucket]
            r[0] = min(r[0], s.offset)
            r[1] = max(r[1], s.offset + s.numel)
            p = nn.Parameter(self.master[s.offset:s.offset + s.numel].view(s.shape), requires_grad=True)
            self.params[s.name] = p
            self.grad_written[s.name] = False
        self.set_expected(())

    # ---- views --------------------------------------------------------------------------------------
    def weights_key(self):
        """changes whenever the fp32 masters may have changed: torch-side writes move the arena's version counter, the native
        optimizer bumps native_epoch"""
        return (self.master.data_ptr(), self.master._version, self.native_epoch)

    def wait_pending(self, bucket: Optional[int] = None) -> None:
        """make the current stream wait for the overlapped optimizer update of ``bucket`` (None: of every bucket still
        pending).  Readers that bypass the view accessors (raw master pointers, ``state_dict()``, ``p.data``) call this."""
        if not self._pending:
            return
        cur = torch.cuda.current_stream(self.device)
        if bucket is None:
            for ev in set(self._pending.values()):
                cur.wait_event(ev)
            self._pending.clear()
            return
        ev = self._pending.pop(bucket, None)
        if ev is not None:
            cur.wait_event(ev)

    def _view(self, arena: torch.Tensor, names: Sequence[str], shape: Optional[Sequence[int]]) -> torch.Tensor:
        s0 = self.slots[names[0]]
        if self._pending:
            self.wait_pending(s0.bucket)
        # the arenas never move and a view is two torch calls (~4 us of the launching thread; MemVLA's step asks for 3,700 of them):
        # each (arena, names, shape) view is built once
        key = (id(arena), names, shape if shape is None else tuple(shape))
        v = self._view_cache.get(key)
        if v is not None:
            return v
        n = 0
        for nm in names:                      # must be packed back to back
            s = self.slots[nm]
            assert s.offset == s0.offset + n, f"{names} are not adjacent in the arena"
            n += s.numel
        v = arena[s0.offset:s0.offset + n]
        v = v.view(tuple(shape)) if shape is not None else v.view(s0.shape) if len(names) == 1 else v
        self._view_cache[key] = v
        return v

    def w(self, *names: str, shape: Optional[Sequence[int]] = None) -> torch.Tensor:
        """compute-dtype weight view (bf16 shadow, or the fp32 master in fp32 mode); several adjacent
        names give the fused matrix"""
        if self.shadow is not None and self.master._version != self._shadow_version:
            self.sync_shadow()            # an in-place torch op (foreign optimizer, load, user code) moved the masters
        arena = self.shadow if self.shadow is not None else self.master
        return self._view(arena, names, shape)

    def w32(self, *names: str, shape: Optional[Sequence[int]] = None) -> torch.Tensor:
        # (the sharded optimizer step gathers the fp32 masters of exactly these buckets after every update: GradReducer.gather_params)
        if self._record_w32:
            self._w32_buckets.add(self.slots[names[0]].bucket)
        return self._view(self.master, names, shape)

    def g(self, *names: str, shape: Optional[Sequence[int]] = None) -> torch.Tensor:
        assert self.grad is not None, "store was finalized with train=False"
        return self._view(self.grad, names, shape)

    # ---- bf16 communication copy of the gradients (data parallel, comm_dtype=bfloat16) -------------------------------
    def enable_grad_mirror(self) -> None:
        """Same layout as the gradient arena, 2 bytes per element (16 GB at 8 B params; the MI355X has 288 GB).  The dW
        GEMM epilogues write it next to the fp32 gradient, the reducer exchanges it in place and AdamW / the grad-norm read
        the averaged copy directly: no cast sweep in either direction."""
        assert self.grad is not None, "store was finalized with train=False"
        if self.gradc is None:
            self.gradc = torch.zeros(self.total, device=self.device, dtype=torch.bfloat16)

    def gc(self, *names: str, shape: Optional[Sequence[int]] = None) -> torch.Tensor:
        assert self.gradc is not None, "enable_grad_mirror() first"
        return self._view(self.gradc, names, shape)

    def mirror_out(self, *names: str, shape: Optional[Sequence[int]] = None) -> Optional[torch.Tensor]:
        """where a gradient-writing kernel should put the bf16 copy of what it writes to g(*names), or None"""
        if self.gradc is None:
            return None
        self._mirrored.update(names)
        return self._view(self.gradc, names, shape)

    # ---- gradient bookkeeping -----------------------------------------------------------------------
    def trainable(self, name: str) -> bool:
        return self.params[name].requires_grad and self.grad is not None

    def accum_flag(self, *names: str) -> bool:
        """True if the gradient slot already holds a value this step (=> kernels must accumulate)."""
        return self.grad_written[names[0]]

    def note_use(self, *names: str) -> None:
        """Forward-side accounting, called by every autograd Function whose backward will write these slots: a
        parameter applied k times in one forward (MemVLA's retrieval / gate blocks run once per sample,
        memvla_arch.py:194-216) receives k gradient contributions, and its bucket is final only after the k-th."""
        for nm in names:
            self._uses[nm] = self._uses.get(nm, 0) + 1

    def mark_written(self, *names: str) -> None:
        """``grad_written`` is sticky for the optimizer step (=> later micro-batches accumulate); the bucket
        countdown restarts every micro-batch so the reducer can fire on the last one.  A slot counts as complete
        when as many backward writes as forward uses (``note_use``) have been enqueued; slots nobody announced
        complete on their first write."""
        for nm in names:
            self.grad_written[nm] = True
            self._micro_touched.add(nm)
            b = self.slots[nm].bucket
            self._bucket_touched[b] = True
            left = self._uses.get(nm, 0)
            if left > 1:
                self._uses[nm] = left - 1          # more consumers of this slot still owe their backward
                continue
            self._uses[nm] = 0
            if nm not in self._micro_written:
                self._micro_written.add(nm)
                self._bucket_pending[b] -= 1
                if self._bucket_pending[b] == 0 and self.on_bucket_ready is not None and not self._bucket_fired[b]:
                    self._bucket_fired[b] = True
                    # the bucket's dW products / bias sums may still be in flight on the side stream.  A hook that runs on a
                    # stream of its own (the norm tracker, the reducer) orders THAT stream behind the side stream
                    # (wait_side); anything else makes the compute stream wait here — which stalls the dX chain for a bias
                    # column sum that nothing downstream of it needs (18 us per decoder layer, profiles/r05_step_timeline.txt)
                    if not (_HOOK_JOINS and getattr(getattr(self.on_bucket_ready, "__self__", None), "joins_side", False)):
                        self.join_wgrad()
                    self.on_bucket_ready(b)

    def wait_side(self, stream) -> None:
        """``stream`` (a hook's own: norm tracker, communication) waits for the gradient work enqueued on the side stream so
        far; the pending flag stays up — the compute stream itself joins at the end of the backward (join_wgrad)"""
        if self._wgrad_pending and self.wgrad_stream is not None and stream is not None:
            stream.wait_stream(self.wgrad_stream)

    def join_wgrad(self) -> None:
        """the current stream waits for the weight-gradient products enqueued on the side stream so far"""
        if self._wgrad_pending and self.wgrad_stream is not None:
            torch.cuda.current_stream(self.device).wait_stream(self.wgrad_stream)
        self._wgrad_pending = False

    def flush_wgrads(self) -> None:
        """weight-gradient products still waiting for a last use that never came (autograd pruned one of the consumers)"""
        if not self._wg_stash and not self._bg_stash:
            return
        from .functional import _bgrad_flush, _wgrad_flush
        for key in list(self._wg_stash):
            for nm in key:
                self._uses[nm] = 1
            _wgrad_flush(self, key)
        for key in list(self._bg_stash):
            for nm in key:
                self._uses[nm] = 1
            _bgrad_flush(self, key)

    def drop_pending_wgrads(self) -> None:
        """forget the weight-gradient products and accumulation pairs of a backward that was ABORTED (out of memory in the
        middle of a coalesced pass, trainer.NativeTrainer.micro_step): nothing is launched, nothing is allocated — the pass is
        re-run from begin_step, whose first writes replace whatever the aborted one left in the gradient arenas"""
        self._wg_stash.clear()
        self._bg_stash.clear()
        self._accum_stash.clear()
        self.join_wgrad()

    def unfired_touched(self) -> List[int]:
        """buckets (highest first = backward order) that received gradient writes this micro-batch but never
        completed their countdown: a frozen / unused slot, or a Function whose backward autograd pruned"""
        return [b for b in reversed(range(len(self.bucket_ranges)))
                if not self._bucket_fired[b] and self._bucket_touched[b]]

    def set_expected(self, exclude: Iterable[str] = ()) -> None:
        """(re)count, per bucket, the slots a backward pass is expected to write: trainable and not in
        ``exclude`` (parameters the path never touches, e.g. lm_head / the unused last CLIP layer)."""
        exclude = set(exclude)
        self._excluded = exclude
        self._bucket_total = [0] * len(self.bucket_ranges)
        for s in self.slots.values():
            if self.params[s.name].requires_grad and s.name not in exclude:
                self._bucket_total[s.bucket] += 1
        self.begin_micro()

    def begin_micro(self) -> None:
        self._micro_written = set()
        self._micro_touched = set()                       # slots that received ANY gradient write this micro-batch
        self._mirrored = set()
        self._uses = {}
        self._bucket_pending = list(self._bucket_total)
        self._bucket_fired = [False] * len(self._bucket_total)
        self._bucket_touched = [False] * len(self._bucket_total)

    def begin_step(self, zero_names: Iterable[str] = ()) -> None:
        """Reset per-step state.  Gradients are produced with beta=0 writes by the GEMM kernels, so
        only slots that are accumulated into (embedding rows, never-written slots) need zeroing."""
        for nm in self.grad_written:
            self.grad_written[nm] = False
        self._accum_stash.clear()
        self.begin_micro()
        for nm in zero_names:
            self.g(nm).zero_()
        self._own_grad_write()
        self._ssq_cursor = 0
        self._ssq_covered = set()

    # ---- sum(g^2) shares produced by the dW products' own epilogues (single-GPU global-norm clip) -------------------
    def sumsq_out(self, names: Sequence[str], M: int, N: int, first_write: Optional[bool] = None) -> Optional[torch.Tensor]:
        """where the product that writes g(*names) as an [M, N] matrix should leave the per-tile partial sums of squares of
        what it writes (``dxa_gemm_desc.sumsq``), or None.  Only when that write is the slot's final value of the step:
        nothing written to it yet and no other consumer of the parameter still owes a gradient (``note_use``)."""
        if not self.epi_sumsq or self.grad is None or not self.grad.is_cuda:
            return None
        # (first_write: the caller knows — the one deferred product of a multiply-used parameter — that nothing has been
        # written to the slot this step although its consumers were already counted down)
        if not self.last_micro:
            return None                                   # an earlier micro-batch: more contributions follow
        # written THIS micro-batch (a slot several kernels add into): not final.  Contributions of EARLIER micro-batches are
        # fine — the product accumulates onto them and the epilogue squares what it stores.
        written = any(nm in self._micro_touched for nm in names) if first_write is None else not first_write
        if written or any(self._uses.get(nm, 0) > 1 for nm in names):
            return None
        from . import kernels as K
        n = K.gemm_sumsq_slots(M, N)
        if self._ssq_buf is None:
            self._ssq_buf = torch.zeros(1 << 21, device=self.device, dtype=torch.float32)
        if self._ssq_cursor + n > self._ssq_buf.numel():
            return None                                   # full: this gradient is read back by the ordinary pass
        out = self._ssq_buf[self._ssq_cursor:self._ssq_cursor + n]
        self._ssq_cursor += n
        self._ssq_covered.update(names)
        return out

    def sumsq_partials(self) -> Optional[torch.Tensor]:
        """the partials handed out since begin_step (their plain sum is the covered slots' sum of squares)"""
        if self._ssq_buf is None or self._ssq_cursor == 0:
            return None
        return self._ssq_buf[:self._ssq_cursor]

    def uncovered_ranges(self, lo: int, hi: int) -> List[Tuple[int, int]]:
        """[lo, hi) of the gradient arena minus the slots whose sum of squares an epilogue already produced this step and
        minus the slots the path never writes (exact zeros); neighbours separated only by alignment padding (zeros) merge"""
        if self._slot_order is None:
            self._slot_order = sorted(self.slots.values(), key=lambda sl: sl.offset)
            self._slot_starts = [sl.offset for sl in self._slot_order]
        import bisect
        i = bisect.bisect_left(self._slot_starts, lo)
        out: List[Tuple[int, int]] = []
        prev_covered = True
        while i < len(self._slot_order) and self._slot_order[i].offset < hi:
            sl = self._slot_order[i]
            i += 1
            skip = sl.name in self._ssq_covered or sl.name in self._excluded or not self.params[sl.name].requires_grad
            if skip:
                prev_covered = True
                continue
            a, b = sl.offset, min(sl.offset + sl.numel, hi)
            if out and not prev_covered:
                out[-1] = (out[-1][0], b)               # only padding in between
            else:
                out.append((a, b))
            prev_covered = False
        return out

    def zero_embed_grad(self, name: str) -> None:
        """make the dense embedding-gradient slice all-zero before this step's first scatter into it: only the rows
        written since it was last all-zero when those are known (``sparse_embed_zero``), else the whole slice"""
        from . import kernels as K
        g = self.g(name)
        dirty = self._embed_dirty.get(name)
        if self.sparse_embed_zero and dirty is not None and g.is_cuda:
            for plan in dirty:
                K.zero_rows(plan, g)
        else:
            g.zero_()
            self._own_grad_write()
        self._embed_dirty[name] = []

    def _own_grad_write(self) -> None:
        """an in-place torch op of this package's own on the gradient arena (external_prelude watches the arena's version
        counter for FOREIGN writes: zero_grad(set_to_none=False), clip_grad_norm_)"""
        if self._grad_version is not None and self.grad is not None:
            self._grad_version = self.grad._version

    def note_embed_rows(self, name: str, plan: torch.Tensor) -> None:
        dirty = self._embed_dirty.get(name)
        if dirty is not None:
            dirty.append(plan)

    def invalidate_embed_tracking(self) -> None:
        """someone else (a gradient all-reduce, user code) wrote the embedding-gradient slice: next zeroing is dense"""
        self._embed_dirty = {}

    def never_written(self) -> List[str]:
        return [nm for nm, wtn in self.grad_written.items() if not wtn and self.params[nm].requires_grad]

    def attach_grads(self) -> None:
        """expose the arena views as ``param.grad`` (HF Trainer / user code compatibility).  Parameters the path never
        writes (``set_expected``'s exclusions: lm_head, the CLIP layer after hidden_states[-2]) keep ``grad = None`` like in
        the reference, so torch.optim skips them (no weight decay, no moments).  With a bf16 gradient arena
        (``bf16_grads``) the weight matrices' gradients live in ``gradc`` as bf16, which torch does not accept as the
        ``.grad`` of an fp32 parameter: that mode exposes no ``p.grad`` at all (read ``store.gc(name)`` / ``store.g(name)``)
        rather than stale fp32 views."""
        if self.grad is None:
            return
        for nm, p in self.params.items():
            if not p.requires_grad:
                continue
            if nm in self._excluded or self.bf16_grads:
                p.grad = None
            else:
                p.grad = self.g(nm)

    def sync_shadow(self) -> None:
        """re-derive the bf16 shadows from the fp32 masters (after load_state_dict / init)"""
        self.wait_pending()
        if self.shadow is not None:
            from . import kernels as K
            K.cast(self.master, self.shadow.dtype, out=self.shadow)
            self._shadow_version = self.master._version

    def external_prelude(self, unused: Iterable[str] = ()) -> None:
        """what a managing trainer does before a training forward, for loops this package does not drive (the model's forward
        pre-hook calls it while ``managed`` is False): HF ``Trainer.training_step`` -> ``optimizer.step()`` ->
        ``model.zero_grad()`` (dexbotic/exp/trainer.py:18-36 inherits exactly that loop).  A NEW optimizer step is recognised by
        gradients that were dropped (``zero_grad(set_to_none=True)``), zeroed / clipped in place (version counter of the
        gradient arena) or reset through the model's own ``zero_grad``; otherwise this forward is a further micro-batch of the
        same step and its gradients accumulate."""
        if self.grad is None:
            return
        unused = set(unused)
        if unused != self._excluded:
            self.set_expected(unused)
        probe = next((self.params[n] for n in self.slots if self.params[n].requires_grad and n not in self._excluded), None)
        dropped = probe is not None and probe.grad is None
        if dropped:
            self.attach_grads()
        if dropped or self._external_fresh or self.grad._version != self._grad_version:
            if self.shadow is not None and self._external_trained:
                # a foreign optimizer has (presumably) moved the fp32 masters since the last training forward.  The arena's
                # version counter does not see every such update — torch.optim.AdamW(fused=True), HF Trainer's default, leaves
                # it untouched — so the bf16 shadows are re-derived at every optimizer-step boundary of an external loop
                # (one 48 GB cast pass at the 7 B size: ~8 ms next to the foreach / fused torch AdamW it follows)
                self.sync_shadow()
            self.begin_step()
            if dropped or self.grad._version != self._grad_version:
                self.invalidate_embed_tracking()          # somebody else wrote the gradient arena: dense re-zero
        else:
            self.begin_micro()
        self._external_fresh = False
        self._external_trained = True
        self._grad_version = self.grad._version

    def external_eval(self) -> None:
        """``model.eval()`` after training under an external loop: the shadows follow the masters once more"""
        if not self.managed and self._external_trained and self.shadow is not None:
            self.sync_shadow()
            self._external_trained = False

    def external_zero_grad(self) -> None:
        """``model.zero_grad()`` of an external loop: gradients stay attached (they are views of the arena, overwritten by the
        next backward's first write); the step boundary is recorded"""
        self._external_fresh = True

class Fp32View:
    """Same interface as ParamStore but ``w()`` hands out the fp32 masters: used by the diffusion action
    head, which the reference keeps in fp32 (cogact_arch.py:133, SURVEY.md App. A dtype notes)."""

    def __init__(self, store: ParamStore):
        self._s = store

    def w(self, *names: str, shape=None):
        return self._s.w32(*names, shape=shape)

    def __getattr__(self, item):
        return getattr(self._s, item)

    def __setattr__(self, item, value):
        # attribute WRITES go to the store too.  Until round 5 they landed on the view: ``st._wgrad_pending = True`` of the
        # action head's side-stream gradient products (functional._wgrad_now / _bgrad) never reached the store, so the head
        # bucket's completion hook ran without joining the side stream — the gradient exchange (and the sum of squares) could
        # read the head's last-written slots before their products had run.  Found by the 2-rank model step
        # (tests/test_zz_dp2_gpu.py; profiles/r05_dp2_race.txt).
        if item == "_s":
            object.__setattr__(self, item, value)
        else:
            setattr(self._s, item, value)


def attach_parameters(root: nn.Module, store: ParamStore, containers: Optional[Dict[str, nn.Module]] = None) -> None:
    """Register every arena view as an nn.Parameter under its dotted name so that ``root.state_dict()``
    has exactly the reference's keys.  Intermediate modules that do not exist yet
    are created as empty containers; existing modules (e.g. the backbone / tower objects) are reused."""
    for name, p in store.params.items():
        parts = name.split(".")
        mod = root
        for i, part in enumerate(parts[:-1]):
            child = mod._modules.get(part)
            if child is None:
                child = ArenaLayerNorm() if ".".join(parts[:i + 1]) in store.layernorm_modules else nn.Module()
                mod.add_module(part, child)
            mod = child
        mod.register_parameter(parts[-1], p)


class ArenaLayerNorm(nn.LayerNorm):
    """parameter container with the TYPE of the reference's module (the arithmetic is libdexbotic_amd's layernorm
    kernels, reached through functional.NormFn / VitBlockFn; calling this object is not part of the native path)"""

    def __init__(self):
        nn.Module.__init__(self)
        self.normalized_shape, self.eps, self.elementwise_affine = (), 1e-5, True

    def forward(self, x):  # pragma: no cover - never on the native path
        raise RuntimeError("ArenaLayerNorm is a parameter container; the native blocks call the layernorm kernels")


# ------------------------------------------------------------------------------------------ optimizer
def no_decay_name(name: str, store: Optional[ParamStore] = None) -> bool:
    """Parameter-group rule of OptimizerConfig._get_optimizer_grouped_parameters (base_exp.py:95-203)
    under the reference's pinned transformers: no weight decay for parameters of nn.LayerNorm modules
    (by module type: ``store.layernorm_modules``) and for every name containing "bias" (RMSNorm weights DO decay)."""
    if "bias" in name:
        return True
    return store is not None and name.rsplit(".", 1)[0] in store.layernorm_modules


@dataclass
class OptimConfig:
    """mirror of OptimizerConfig (base_exp.py:64-93) + the trainer link (trainer.py:88-124)"""
    base_lr: float = 2e-5
    weight_decay: float = 0.0
    adam_beta1: float = 0.9
    adam_beta2: float = 0.999
    adam_epsilon: float = 1e-8
    max_grad_norm: Optional[float] = 1.0
    mm_projector_lr: Optional[float] = None
    mm_vision_lr: Optional[float] = None
    action_head_lr: Optional[float] = None
    chunk: int = 32768


class FusedAdamW:
    """One-launch AdamW over the arena + device-side global-norm clipping."""

    def __init__(self, store: ParamStore, cfg: OptimConfig, prefixes: Dict[str, str] | None = None,
                 exclude: Iterable[str] = (), overlap: bool = False, segment_elems: int = 48 << 20,
                 groups: Optional[List[dict]] = None, ranges: Optional[Sequence[Tuple[int, int]]] = None):
        """``ranges``: sharded optimizer state — sorted, disjoint arena ranges [a, b) this rank owns (ShardPlan.owned): only
        parameters inside them are updated here, and m / v exist for them alone, packed back to back (``dxa_adamw_desc.
        chunk_mv_start``) — 8 bytes per OWNED parameter instead of 8 per parameter.  None: every trainable parameter.
        ``exclude``: parameters that never receive a gradient (torch.optim skips ``grad is None``
        parameters entirely — no weight decay either).
        ``overlap``: run the update on a side HIP stream, cut along bucket boundaries into segments of >= ``segment_elems``
        parameters in FORWARD order, one event per segment (``ParamStore._pending``): the next step's forward waits
        per bucket instead of for the whole 224 GB sweep — the update is HBM-bound, the forward MFMA-bound.
        ``groups``: explicit parameter groups ``[{"names": [...]}, ...]`` (<= 8; exp.trainer.ArenaAdamW passes the groups the
        reference's OptimizerConfig built) instead of the name rule below; learning rate and weight decay of group i are then
        given per step (``step(lrs=, wds=)``).  Parameters in no group are not updated."""
        from . import kernels as K  # noqa: F401  (fail early if the library is missing)
        self.store, self.cfg = store, cfg
        self.overlap = bool(overlap) and store.device.type == "cuda"
        assert not (self.overlap and ranges is not None), "the overlapped update and the sharded optimizer state exclude each other"
        exclude = set(exclude)
        dev = store.device
        self.ranges = None if ranges is None else [(int(a), int(b)) for a, b in ranges if b > a]
        if self.ranges is None:
            self.m = torch.zeros_like(store.master)
            self.v = torch.zeros_like(store.master)
        else:
            assert all(self.ranges[i][1] <= self.ranges[i + 1][0] for i in range(len(self.ranges) - 1)), "ranges must be sorted and disjoint"
            # packed offset of every owned range inside m / v: ranges start on 16-byte boundaries there as they do in the arena
            # (adamw_k takes its 16-byte path only when both offsets allow it)
            self._range_base, owned = [], 0
            for a, b in self.ranges:
                self._range_base.append(owned)
                owned += (b - a + 3) // 4 * 4
            self.m = torch.zeros(max(owned, 1), device=dev, dtype=torch.float32)
            self.v = torch.zeros(max(owned, 1), device=dev, dtype=torch.float32)
        self.step_count = 0
        prefixes = prefixes or {"mm_projector": "mm_projector", "mm_vision": "mm_vision", "action_head": "action_head"}
        # groups: (lr_key, decay?) -> index  (<= 8 groups, as the reference builds them)
        self.group_keys: List[Tuple[str, bool]] = []
        self.group_of: Dict[str, Tuple[str, bool]] = {}      # parameter name -> (lr key, weight-decayed?)
        cs, cl, cg, cst, cms = [], [], [], [], []
        import bisect as _bis
        r_lo = [a for a, _ in self.ranges] if self.ranges is not None else None
        r_base = self._range_base if self.ranges is not None else []

        def pieces(c0: int, c1: int):
            """[c0, c1) cut to the owned ranges -> (start, length, packed moment offset)"""
            if self.ranges is None:
                yield c0, c1 - c0, None
                return
            k = max(0, _bis.bisect_right(r_lo, c0) - 1)
            while k < len(self.ranges) and self.ranges[k][0] < c1:
                a, b = self.ranges[k]
                x0, x1 = max(a, c0), min(b, c1)
                if x1 > x0:
                    yield x0, x1 - x0, r_base[k] + (x0 - a)
                k += 1
        # token-embedding tables receive gradient in <= B * S_text rows per step: their chunks start in state 1 = "never saw a
        # non-zero gradient" and the kernel leaves such a chunk alone while its gradient is all-zero and its weight decay is 0 —
        # exactly what AdamW computes for it (dxa_adamw_desc.chunk_state).  DXA_ADAMW_SPARSE=0: every chunk ordinary.
        sparse_tables = __import__("os").environ.get("DXA_ADAMW_SPARSE", "1") != "0"
        explicit = None
        self._explicit_defaults: Optional[List[Tuple[Optional[float], Optional[float]]]] = None
        if groups is not None:
            explicit = {nm: gi for gi, g in enumerate(groups) for nm in g["names"]}
            self.group_keys = [(f"group{gi}", True) for gi in range(len(groups))]
            # per-group defaults for callers that step without lrs / wds: an entry may carry "lr" / "weight_decay" (the
            # reference's groups do, base_exp.py:95-203); a group without them refuses to be stepped on guessed values
            self._explicit_defaults = [(g.get("lr"), g.get("weight_decay")) for g in groups]
        for s in sorted(store.slots.values(), key=lambda s: s.offset):
            if not store.params[s.name].requires_grad or s.name in exclude:
                continue
            if explicit is not None:
                if s.name not in explicit:
                    continue
                key = self.group_keys[explicit[s.name]]
            else:
                lr_key = "base"
                if cfg.mm_projector_lr is not None and prefixes["mm_projector"] in s.name:
                    lr_key = "mm_projector"
                elif cfg.mm_vision_lr is not None and prefixes["mm_vision"] in s.name:
                    lr_key = "mm_vision"
                elif cfg.action_head_lr is not None and prefixes["action_head"] in s.name:
                    lr_key = "action_head"
                key = (lr_key, not no_decay_name(s.name, store))
            self.group_of[s.name] = key
            if key not in self.group_keys:
                self.group_keys.append(key)
            gi = self.group_keys.index(key)
            o = 0
            while o < s.numel:
                ln = min(cfg.chunk, s.numel - o)
                for x0, xl, mo in pieces(s.offset + o, s.offset + o + ln):
                    cs.append(x0)
                    cl.append(xl)
                    cg.append(gi)
                    cst.append(1 if sparse_tables and s.name.endswith("embed_tokens.weight") else 0)
                    cms.append(mo)
                o += ln
        assert len(self.group_keys) <= 8
        self.chunk_mv_start = torch.tensor(cms, dtype=torch.int64, device=dev) if self.ranges is not None else None
        # segments for the overlapped update: [first chunk, one past the last chunk, buckets covered]
        self.segments: List[Tuple[int, int, List[int]]] = []
        self.stream = None
        if self.overlap:
            import bisect
            prio = int(__import__("os").environ.get("DXA_OPT_STREAM_PRIO", "0"))
            self.stream = torch.cuda.Stream(device=dev, priority=prio)
            i0, buckets, elems = 0, [], 0
            for b, (lo, hi) in enumerate(store.bucket_ranges):
                if hi <= lo:
                    continue
                buckets.append(b)
                i1 = bisect.bisect_left(cs, hi)
                elems = sum(cl[i0:i1])
                if elems >= segment_elems:
                    self.segments.append((i0, i1, buckets))
                    i0, buckets = i1, []
            if buckets or i0 < len(cs):
                self.segments.append((i0, len(cs), buckets))
            self.segments = [sg for sg in self.segments if sg[1] > sg[0] or sg[2]]
            self._events = [torch.cuda.Event() for _ in self.segments]
        self.chunk_start = torch.tensor(cs, dtype=torch.int64, device=dev)
        self.chunk_len = torch.tensor(cl, dtype=torch.int32, device=dev)
        self.chunk_grp = torch.tensor(cg, dtype=torch.int32, device=dev)
        self.chunk_state = torch.tensor(cst, dtype=torch.uint8, device=dev) if any(cst) and dev.type == "cuda" else None
        self.sumsq = torch.zeros(1, device=dev, dtype=torch.float32)
        self.norm = torch.zeros(1, device=dev, dtype=torch.float32)
        self.coef = torch.ones(1, device=dev, dtype=torch.float32)
        self.scratch = torch.empty(4096, device=dev, dtype=torch.float64)

    def _lrs_wds(self, lr_scale: float):
        c = self.cfg
        if self._explicit_defaults is not None:
            if any(lr is None or wd is None for lr, wd in self._explicit_defaults):
                raise ValueError("FusedAdamW was built from explicit parameter groups without 'lr' / 'weight_decay' entries: "
                                 "pass lrs= and wds= to step() (exp.trainer.ArenaAdamW does), base_lr / the name rule do not "
                                 "apply to explicit groups")
            return ([float(lr) * lr_scale for lr, _ in self._explicit_defaults],
                    [float(wd) for _, wd in self._explicit_defaults])
        lr_of = {"base": c.base_lr, "mm_projector": c.mm_projector_lr, "mm_vision": c.mm_vision_lr,
                 "action_head": c.action_head_lr}
        lrs = [lr_of.get(k, c.base_lr) * lr_scale for k, _ in self.group_keys]
        wds = [c.weight_decay if dec else 0.0 for _, dec in self.group_keys]
        return lrs, wds

    def step(self, lr_scale: float = 1.0, sumsq: Optional[torch.Tensor] = None,
             grads: Optional[torch.Tensor] = None, lrs: Optional[Sequence[float]] = None,
             wds: Optional[Sequence[float]] = None, grad_scale: float = 1.0) -> None:
        """``sumsq``: device scalar holding sum(g^2) over the arena if someone already accumulated it (GradNormTracker
        does, bucket by bucket under the backward); otherwise one pass over the gradient arena computes it here.
        ``grads``: arena to read the gradients from (default the fp32 gradient arena; the averaged bf16 communication
        copy under bf16 data parallelism).  ``grad_scale``: the arena holds gradient / grad_scale (data parallelism with SUM
        collectives: world x the mean, grad_scale = 1 / world) — the norm is the scaled gradient's and the factor rides in the
        clip coefficient, so the update is exactly the mean gradient's."""
        from . import kernels as K
        st, c = self.store, self.cfg
        grads = st.grad if grads is None else grads
        self.step_count += 1
        if self.overlap:
            st.wait_pending()                      # a previous update nobody consumed still reads the clip coefficient
        clip = None
        if c.max_grad_norm is not None:
            if sumsq is None:
                assert self.ranges is None, "sharded optimizer state: the caller supplies the all-reduced sum of squares"
                K.sumsq(grads, self.sumsq, self.scratch)
                sumsq = self.sumsq
            K.clip_coef(sumsq, float(c.max_grad_norm), self.norm, self.coef, grad_scale=float(grad_scale))
            clip = self.coef
        elif grad_scale != 1.0:
            self.coef.fill_(float(grad_scale))
            clip = self.coef
        if lrs is None or wds is None:
            d_lrs, d_wds = self._lrs_wds(lr_scale)
            lrs = d_lrs if lrs is None else lrs
            wds = d_wds if wds is None else wds
        lrs, wds = [float(x) for x in lrs], [float(x) for x in wds]
        assert len(lrs) == len(wds) == len(self.group_keys)
        st.native_epoch += 1                       # the masters move under raw pointers: derived copies (packed DiT weights) are stale
        if not self.overlap:
            K.adamw(st.master, grads, self.m, self.v, st.shadow, self.chunk_start, self.chunk_len, self.chunk_grp,
                    lrs, wds, c.adam_beta1, c.adam_beta2, c.adam_epsilon, self.step_count, clip=clip, chunk_state=self.chunk_state,
                    chunk_mv_start=self.chunk_mv_start)
            return
        cur = torch.cuda.current_stream(st.device)
        self.stream.wait_stream(cur)               # gradients, sum(g^2) and the clip coefficient are final on `cur`
        with torch.cuda.stream(self.stream):
            for (i0, i1, buckets), ev in zip(self.segments, self._events):
                if i1 > i0:
                    K.adamw(st.master, grads, self.m, self.v, st.shadow, self.chunk_start[i0:i1], self.chunk_len[i0:i1],
                            self.chunk_grp[i0:i1], lrs, wds, c.adam_beta1, c.adam_beta2, c.adam_epsilon, self.step_count,
                            clip=clip, chunk_state=None if self.chunk_state is None else self.chunk_state[i0:i1])
                ev.record(self.stream)
                for b in buckets:
                    st._pending[b] = ev

    def synchronize(self) -> None:
        """the current stream waits for an overlapped update still in flight"""
        self.store.wait_pending()

    def moments_replaced(self) -> None:
        """m / v were written from outside (an optimizer state dict was loaded): no chunk may be assumed to hold zero moments"""
        if self.chunk_state is not None:
            self.chunk_state.zero_()

    def load_moments(self, m: torch.Tensor, v: torch.Tensor, step: int) -> None:
        """THE way to write the moment arenas from outside (resume, tests copying a state in): the sparse-table skip of
        ``adamw_k`` (chunk_state == 1: "this embedding chunk never saw a non-zero gradient, so m = v = 0") is an invariant of
        moments this object produced itself — every external write resets it.  ``m`` / ``v`` may be the arenas themselves
        (state dict round trip in place)."""
        self.store.wait_pending()
        if self.ranges is not None and m.numel() == self.store.total and v.numel() == self.store.total:
            m, v = self.pack_moments(m), self.pack_moments(v)     # full arenas (a gathered checkpoint): keep the own shard
        assert m.shape == self.m.shape and v.shape == self.v.shape, \
            "moments of another layout (a sharded optimizer state loads full arenas or what the same world size / rank saved)"
        if m.data_ptr() != self.m.data_ptr():
            self.m.copy_(m)
        if v.data_ptr() != self.v.data_ptr():
            self.v.copy_(v)
        self.step_count = int(step)
        self.moments_replaced()


    def pack_moments(self, full: torch.Tensor) -> torch.Tensor:
        """arena-shaped moments -> this rank's packed shard"""
        full = full.to(self.m.device).reshape(-1)
        out = torch.zeros_like(self.m)
        for (a, b), o in zip(self.ranges, self._range_base):
            out[o:o + (b - a)] = full[a:b]
        return out

    def full_moments(self, reducer: Optional["GradReducer"] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """arena-shaped (m, v): the arenas themselves unsharded; sharded, two fresh arenas holding every rank's shard (all ranks
        call it: the shards travel by the reducer's all-gathers) — what a checkpoint that can be resumed on another world size
        stores.  64 GB of scratch at the 8 B size, freed by the caller."""
        if self.ranges is None:
            return self.m, self.v
        out = []
        for packed in (self.m, self.v):
            full = torch.zeros(self.store.total, device=packed.device, dtype=torch.float32)
            for (a, b), o in zip(self.ranges, self._range_base):
                full[a:b] = packed[o:o + (b - a)]
            out.append(full)
        if reducer is not None and reducer.plan is not None and reducer.world > 1:
            reducer._gather_slices(lambda sl: out)
            self.store.wait_pending()
        return out[0], out[1]


def cosine_lr_scale(step: int, total_steps: int, warmup_steps: int = 0) -> float:
    """HF get_cosine_schedule_with_warmup (lr_scheduler_type="cosine", trainer.py:88-124)."""
    if step < warmup_steps:
        return step / max(1, warmup_steps)
    prog = (step - warmup_steps) / max(1, total_steps - warmup_steps)
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * prog)))


# --------------------------------------------------------------------------- overlapped gradient norm
class GradNormTracker:
    """sum(g^2) for the global-norm clip, accumulated bucket by bucket on a side HIP stream while the backward is
    still running (the pass is HBM-bound, the backward MFMA-bound), instead of one 32 GB sweep in front of AdamW.
    Buckets are folded in as they become final: straight from ``ParamStore.on_bucket_ready`` on one GPU, after the
    bucket's all-reduce on the communication stream under data parallelism (``GradReducer.after_reduce``).  The
    accumulation order is the (deterministic) bucket order; slots no kernel writes hold zeros and add nothing."""

    def __init__(self, store: ParamStore, min_bytes: int = 256 << 20):
        self.store = store
        self.min_bytes = min_bytes
        self.src: Optional[torch.Tensor] = None      # arena the norm is taken over (default: the fp32 gradient arena)
        dev = store.device
        self.acc = torch.zeros(1, device=dev, dtype=torch.float32)
        self.scratch = torch.empty(4096, device=dev, dtype=torch.float64)
        self.stream = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None
        self.joins_side = self.stream is not None    # bucket_ready orders its own stream behind the gradient side stream
        self._lo: Optional[int] = None
        self._hi: Optional[int] = None
        self._range_cache: Dict[tuple, tuple] = {}
        self._small: List[Tuple[int, int]] = []      # short uncovered slices of this step, summed in finish()

    BIG = 1 << 20     # uncovered slices at least this long get the two-stage pass of their own, shorter ones share a launch
    CHUNK = 1 << 13   # ... cut into chunks of this many elements, one workgroup each

    def begin(self) -> None:
        self.acc.zero_()                       # on the compute stream; every fold waits for that stream first
        self._lo = self._hi = None
        self._small = []

    def fold(self, lo: int, hi: int, after=None) -> None:
        """acc += sum(grad[lo:hi]^2) on the tracker's side stream, ordered after everything enqueued so far on ``after``
        (default: the compute stream; the communication stream when the slice has just been all-reduced there — the
        collectives that follow on that stream are not held up by the reduction)"""
        from . import kernels as K
        if hi <= lo:
            return
        src = self.store.grad if self.src is None else self.src
        if self.stream is None:
            K.sumsq(src[lo:hi], self.acc, self.scratch, accumulate=True)
            return
        self.stream.wait_stream(after if after is not None else torch.cuda.current_stream())
        self.store.wait_side(self.stream)
        with torch.cuda.stream(self.stream):
            if not self.store.epi_sumsq:
                K.sumsq(src[lo:hi], self.acc, self.scratch, accumulate=True)
                return
            # the dW products already left their share in the store's partial buffer: read back only what they do not cover.
            # Short slices (norm weights, biases, position embeddings: kilobytes each) are only NOTED here and summed by ONE launch
            # in finish() — round 4 launched a pair of kernels per bucket for them (69 + 35 launches per step)
            for a, b in self.store.uncovered_ranges(lo, hi):
                if b - a >= self.BIG:
                    K.sumsq(src[a:b], self.acc, self.scratch, accumulate=True)
                else:
                    for c in range(a, b, self.CHUNK):                   # one workgroup per chunk
                        self._small.append((c, min(self.CHUNK, b - c)))

    def _fold_small(self) -> None:
        """the short slices noted by fold(), one launch per 4096 of them (on the tracker's stream; every slice is final: the
        caller has joined the gradient side stream)"""
        from . import kernels as K
        if not self._small:
            return
        src = self.store.grad if self.src is None else self.src
        small, self._small = self._small, []
        for i in range(0, len(small), 4096):
            chunk = tuple(small[i:i + 4096])
            dev = self._range_cache.get(chunk)
            if dev is None:
                dev = (torch.tensor([c[0] for c in chunk], dtype=torch.int64, device=src.device),
                       torch.tensor([c[1] for c in chunk], dtype=torch.int64, device=src.device))
                self._range_cache[chunk] = dev
            K.sumsq_ranges(src, dev[0], dev[1], self.acc, self.scratch, accumulate=True)

    def bucket_ready(self, b: int) -> None:
        lo, hi = self.store.bucket_ranges[b]
        if hi <= lo:
            return
        if self._lo is None:
            self._lo, self._hi = lo, hi
        elif hi == self._lo or abs(self._lo - hi) < ALIGN:
            self._lo = lo
        elif lo == self._hi or abs(lo - self._hi) < ALIGN:
            self._hi = hi
        else:
            self.fold(self._lo, self._hi)
            self._lo, self._hi = lo, hi
        if (self._hi - self._lo) * 4 >= self.min_bytes:
            self.fold(self._lo, self._hi)
            self._lo = self._hi = None

    def finish(self, fire_unfired: bool = True) -> torch.Tensor:
        """fold what is pending (and, on one GPU, the buckets an unused slot kept from firing), then make the compute
        stream wait for the side stream; returns the device scalar"""
        st = self.store
        if fire_unfired:
            for b in st.unfired_touched():
                st._bucket_fired[b] = True
                self.bucket_ready(b)
        if self._lo is not None:
            self.fold(self._lo, self._hi)
            self._lo = self._hi = None
        if self.stream is not None:
            part = st.sumsq_partials() if st.epi_sumsq else None
            if part is not None or self._small:
                from . import kernels as K
                self.stream.wait_stream(torch.cuda.current_stream())        # the last products' epilogues, the last short slices
                with torch.cuda.stream(self.stream):
                    self._fold_small()
                    if part is not None:
                        K.sum_f32(part, self.acc, accumulate=True)
            torch.cuda.current_stream().wait_stream(self.stream)
        else:
            self._fold_small()
        return self.acc


# ---------------------------------------------------------------------------- sharded optimizer step
class ShardPlan:
    """Static partition of the arena for the SHARDED optimizer step — the counterpart of the reference's default DeepSpeed
    config (dexbotic/exp/base_exp.py:229 ``deepspeed="./script/deepspeed/zero3.json"``, script/deepspeed/zero3.json:17-25
    ``"stage": 3, "overlap_comm": true``; the trainer side is dexbotic/exp/trainer.py:145-189): optimizer state and update are
    partitioned over the data-parallel ranks.  Here the weights stay replicated (bf16 shadows: 16 GB of 288) and the step is

        reduce-scatter(gradient slice) -> sum(g^2) over the own shard + ONE scalar all-reduce -> adamw_k over the own shard
        -> all-gather of the updated bf16 SHADOW shard (fp32 masters only for buckets some forward reads in fp32: the action head)

    instead of reduce-scatter + all-gather of the GRADIENT and a full adamw_k on every rank: the same reduce-scatter, an
    all-gather of half the bytes (bf16 weights instead of fp32 gradients; equal with a bf16 exchange), and 1 / world of the
    34 ms, HBM-bound update (14 % of the 1-GPU step).  fp32 masters of the other ranks' shards go stale on a rank
    (GradReducer.gather_masters() brings them up to date for a checkpoint); m / v exist for the own shard only.

    ``slices``: runs of adjacent exchanged buckets of at least ``min_bucket_bytes`` in ARENA (= forward) order — fixed, unlike the
    replicated path's merging by completion order, because ownership must not depend on when a backward fires its buckets.
    Slice [lo, hi) with n = hi - lo: per = n // world rounded down to 16 elements (32-byte shard starts in bf16), rank r owns
    [lo + r per, lo + (r + 1) per); the tail [lo + world per, hi) (< 16 world elements + the remainder) is all-reduced and
    updated by EVERY rank identically (replicated: cheaper than a ragged collective).  per < 64: the whole slice is a tail."""

    def __init__(self, store: "ParamStore", world: int, rank: int, skip_buckets: Iterable[int] = (),
                 min_bucket_bytes: int = 256 << 20):
        assert world >= 1 and 0 <= rank < world
        self.world, self.rank = int(world), int(rank)
        skip = set(skip_buckets)
        self.slices: List[dict] = []
        cur = None
        for b, (lo, hi) in enumerate(store.bucket_ranges):
            if hi <= lo:
                continue
            if b in skip:                                   # never exchanged, never updated: a hole between slices
                if cur is not None:
                    self.slices.append(cur)
                    cur = None
                continue
            if cur is not None and 0 <= lo - cur["hi"] < ALIGN:
                cur["hi"] = hi
                cur["buckets"].append(b)
            else:
                if cur is not None:
                    self.slices.append(cur)
                cur = {"lo": lo, "hi": hi, "buckets": [b]}
            if (cur["hi"] - cur["lo"]) * 4 >= min_bucket_bytes:
                self.slices.append(cur)
                cur = None
        if cur is not None:
            self.slices.append(cur)
        self.slice_of: Dict[int, int] = {}
        for i, sl in enumerate(self.slices):
            n = sl["hi"] - sl["lo"]
            per = n // self.world
            per = per - per % 16 if per >= 64 else 0
            sl["per"], sl["body"] = per, per * self.world
            for b in sl["buckets"]:
                self.slice_of[b] = i

    def shard(self, i: int, rank: Optional[int] = None) -> Tuple[int, int]:
        sl = self.slices[i]
        r = self.rank if rank is None else rank
        return sl["lo"] + r * sl["per"], sl["lo"] + (r + 1) * sl["per"]

    def tail(self, i: int) -> Tuple[int, int]:
        sl = self.slices[i]
        return sl["lo"] + sl["body"], sl["hi"]

    def owned(self, rank: Optional[int] = None) -> List[Tuple[int, int]]:
        """what rank ``rank`` updates: its shard of every slice and every (replicated) tail; sorted, adjacent ranges merged"""
        out: List[Tuple[int, int]] = []
        for i in range(len(self.slices)):
            for a, b in (self.shard(i, rank), self.tail(i)):
                if b > a:
                    if out and out[-1][1] == a:
                        out[-1] = (out[-1][0], b)
                    else:
                        out.append((a, b))
        return sorted(out)

    def describe(self) -> dict:
        own = sum(b - a for a, b in self.owned())
        tot = sum(sl["hi"] - sl["lo"] for sl in self.slices)
        return {"world": self.world, "rank": self.rank, "slices": len(self.slices), "owned_elements": own, "exchanged_elements": tot,
                "replicated_tail_elements": sum(sl["hi"] - sl["lo"] - sl["body"] for sl in self.slices)}


# ---------------------------------------------------------------------------------------- DP reducer
class GradReducer:
    """Data-parallel gradient averaging over RCCL (torch.distributed backend "nccl" on ROCm; "gloo" in the CPU tests).

    Buckets are contiguous slices of the gradient arena in reverse forward order (DiT head -> LLM layers 27..0 ->
    projector -> ViT); a bucket is exchanged in place on a side HIP stream as soon as every slot in it has received its
    last gradient write, so the communication of block i overlaps the backward compute of blocks < i.  Buckets are merged
    up to ``min_bucket_bytes`` per collective; parameters that never receive a gradient (lm_head, the unused last CLIP
    layer) are never sent (the reference papers over them with ddp_find_unused_parameters=True, trainer.py:121).

    ``algo="rs_ag"`` (default): reduce-scatter(AVG) into this rank's 1/world shard of the merged slice, then all-gather
    of the shards, both in place — the two halves of a ring all-reduce issued separately, so that every rank sources
    and sinks 1/world of every slice over each of its xGMI links (point-to-point fabric, 7 links per GPU).  Lengths not
    divisible by the world size: the divisible prefix goes through RS+AG, the < world-element tail through one tiny
    all-reduce.  ``algo="allreduce"``: one ``all_reduce(AVG)`` per merged slice (RCCL picks the algorithm).

    ``comm_dtype=bfloat16``: what is exchanged is the bf16 communication copy of the gradients (ParamStore.gradc; what
    the reference's DeepSpeed bf16 run reduces, script/deepspeed/zero2.json): half the xGMI bytes.  The dW GEMM
    epilogues write that copy next to the fp32 gradient; the few slots written by other kernels (norm weights, biases,
    embeddings) are cast on the communication stream just before their slice is sent; the optimizer and the grad-norm
    read the averaged copy directly (``result_arena``), so there is no cast sweep in either direction.
    """

    def __init__(self, store: ParamStore, group=None, min_bucket_bytes: int = 256 << 20,
                 skip: Iterable[str] = (), force: bool = False, comm_dtype: torch.dtype = torch.float32,
                 algo: str = "rs_ag", local_only: bool = False, native_avg_world1: bool = False,
                 reduce_op: str = "sum", shard: bool = False, emulate_world: int = 0):
        """``shard``: the sharded optimizer step (ShardPlan): gradients are reduce-SCATTERED over fixed slices — no gradient
        all-gather — ``after_reduce`` sees this rank's shard (and, on rank 0, the replicated tails) only, and gather_params() /
        gather_masters() bring the updated weights back.  ``emulate_world`` = n at world size 1 (``force``): ownership as rank 0 of n
        ranks while the collectives run over the whole slices at world size 1 — the timing of an n-rank step's local work
        on one GPU (bench.py ``dp8_emulated_ms_per_step``); the parameters of the other n - 1 shards are simply not updated."""
        import torch.distributed as dist
        self.dist = dist
        # reduce_op = "sum" (default since round 5): the collectives ADD the ranks' gradients and the 1 / world of the mean is
        # folded into the clip coefficient AdamW multiplies every gradient by anyway (``grad_scale``; FusedAdamW.step,
        # dxa_clip_coef_scaled): the arena then holds world x the mean gradient, the reported norm and the update are the
        # mean's.  RCCL's AVG is a pre-multiply inside the reduction kernel; next to the backward's GEMM grids the AVG sequence
        # measured 274 ms per step against 252 for SUM on one MI355X (profiles/r04_reducer_contention.json).  "avg": the
        # collectives average (native AVG on RCCL, SUM + divide on gloo) and the arena holds the mean.
        assert reduce_op in ("sum", "avg")
        self.reduce_op = reduce_op
        self.force = force or local_only   # run the bucket pipeline even at world size 1 (exercises the RCCL path)
        # local_only: no process group at all — the per-bucket pipeline (bf16 copy of the slots no epilogue wrote, sum of squares
        # on the side stream) without any collective: the single-GPU bf16-gradient step
        self.local_only = local_only
        # world size 1 normally asks RCCL for SUM (the mean of one rank is the identity; RCCL's one-rank AVG is a separate
        # pre-multiply pass).  native_avg_world1=True takes the EXACT call sequence every N > 1 run takes instead — in-place
        # reduce_scatter_tensor(AVG) into this rank's shard + all_gather_into_tensor, shard alignment, all-reduced tail — so
        # that path executes on hardware with one GPU (tests/test_zz_dp_gpu.py)
        self.native_avg_world1 = native_avg_world1
        self.time_comm = False                          # bench.py: record the per-step communication window
        self._t0 = None
        self._windows: List[tuple] = []
        assert comm_dtype in (torch.float32, torch.bfloat16)
        assert algo in ("rs_ag", "allreduce")
        self.comm_dtype, self.algo = comm_dtype, algo
        self.store = store
        self.group = group
        init = dist.is_available() and dist.is_initialized() and not local_only
        self.world = dist.get_world_size(group) if init else 1
        self.rank = dist.get_rank(group) if init else 0
        self.backend = str(dist.get_backend(group)) if init else "none"
        # what the optimizer multiplies the exchanged gradients by (1.0: the arena already holds the mean)
        self.grad_scale = 1.0 / self.world if (reduce_op == "sum" and self.world > 1) else 1.0
        # a backend without device collectives (gloo: the 2-rank MODEL step on one MI355X, tests/test_dp2_gpu.py) gets the
        # slices through pinned host memory; RCCL ("nccl") takes the device pointers
        self.stage_host = store.device.type == "cuda" and self.backend != "nccl"
        self.min_bucket_bytes = min_bucket_bytes
        self.skip_buckets = set()
        self.comm_stream = torch.cuda.Stream(device=store.device) if store.device.type == "cuda" else None
        self.joins_side = self.comm_stream is not None   # _flush orders the communication stream behind the gradient side stream
        self._pending_lo: Optional[int] = None
        self._pending_hi: Optional[int] = None
        self.after_reduce = None    # callable(lo, hi, comm_stream): called once a slice's exchange is enqueued
        self.bytes_reduced = 0
        self.bytes_gathered = 0                         # sharded step: bytes of updated weights all-gathered
        self.collectives = 0
        if comm_dtype == torch.bfloat16:
            store.enable_grad_mirror()
        skip = set(skip)
        # buckets whose every slot is skipped (never gets a gradient) are not communicated
        by_bucket: Dict[int, List[str]] = {}
        for s in store.slots.values():
            by_bucket.setdefault(s.bucket, []).append(s.name)
        for b, names in by_bucket.items():
            if all((n in skip) or (not store.params[n].requires_grad) for n in names):
                self.skip_buckets.add(b)
        self._slots = sorted(store.slots.values(), key=lambda s: s.offset)
        self._offsets = [s.offset for s in self._slots]
        store.on_bucket_ready = self.bucket_ready
        self.plan: Optional[ShardPlan] = None
        self.emulate_world = int(emulate_world) if (self.world == 1 and emulate_world and emulate_world > 1) else 0
        if shard and not local_only and (self.world > 1 or self.force):
            assert algo == "rs_ag", "the sharded optimizer step rides on reduce-scatter + all-gather"
            self.plan = ShardPlan(store, self.emulate_world or self.world, 0 if self.emulate_world else self.rank,
                                  self.skip_buckets, min_bucket_bytes)
            self._slice_left = [len(sl["buckets"]) for sl in self.plan.slices]
            self._slice_done = [False] * len(self.plan.slices)
            self._gather_events: List[Optional["torch.cuda.Event"]] = [None] * len(self.plan.slices)

    @property
    def result_arena(self) -> torch.Tensor:
        """the arena that holds the averaged gradients once finish() has returned"""
        return self.store.gradc if self.comm_dtype == torch.bfloat16 else self.store.grad

    # ---- bf16 copy of the slots no GEMM epilogue mirrored ------------------------------------------------------------
    def _fill_mirror(self, lo: int, hi: int) -> None:
        import bisect
        st = self.store
        i = max(0, bisect.bisect_right(self._offsets, lo) - 1)
        run_lo = run_hi = None
        while i < len(self._slots) and self._slots[i].offset < hi:
            s = self._slots[i]
            i += 1
            if s.offset + s.numel <= lo:
                continue
            need = st.grad_written.get(s.name, False) and s.name not in st._mirrored
            if need and run_hi is not None and s.offset - run_hi < ALIGN:
                run_hi = s.offset + s.numel                      # alignment gap rides along (zeros on both sides)
                continue
            if run_lo is not None:
                self._cast_run(run_lo, run_hi)
                run_lo = run_hi = None
            if need:
                run_lo, run_hi = s.offset, s.offset + s.numel
        if run_lo is not None:
            self._cast_run(run_lo, run_hi)

    def _cast_run(self, lo: int, hi: int) -> None:
        st = self.store
        if st.grad.is_cuda:
            from . import kernels as K
            K.cast(st.grad[lo:hi], torch.bfloat16, out=st.gradc[lo:hi])
        else:
            st.gradc[lo:hi].copy_(st.grad[lo:hi])

    # ---- the exchange ------------------------------------------------------------------------------------------------
    def _all_reduce(self, t: torch.Tensor, op) -> None:
        if self.stage_host:
            h = t.to("cpu")                                     # (synchronises the communication stream: test path only)
            self.dist.all_reduce(h, op=op, group=self.group)
            t.copy_(h, non_blocking=False)
        else:
            self.dist.all_reduce(t, op=op, group=self.group)

    def _reduce_scatter(self, shard: torch.Tensor, body: torch.Tensor, op) -> None:
        if self.stage_host:
            hb = body.to("cpu")
            hs = torch.empty(shard.shape, dtype=shard.dtype)
            self.dist.reduce_scatter_tensor(hs, hb, op=op, group=self.group)
            shard.copy_(hs, non_blocking=False)
        else:
            self.dist.reduce_scatter_tensor(shard, body, op=op, group=self.group)

    def _all_gather(self, body: torch.Tensor, shard: torch.Tensor) -> None:
        if self.stage_host:
            hb = torch.empty(body.shape, dtype=body.dtype)
            self.dist.all_gather_into_tensor(hb, shard.to("cpu"), group=self.group)
            body.copy_(hb, non_blocking=False)
        else:
            self.dist.all_gather_into_tensor(body, shard, group=self.group)

    def _avg(self, t: torch.Tensor, native_avg: bool) -> None:
        d = self.dist
        if self.reduce_op == "sum" and not (self.world == 1 and self.native_avg_world1):
            self._all_reduce(t, d.ReduceOp.SUM)          # (--native-avg at one rank: AVG in body AND tail, the N > 1 AVG sequence)
        elif native_avg:
            self._all_reduce(t, d.ReduceOp.AVG)
        else:                                                    # gloo has no AVG
            self._all_reduce(t, d.ReduceOp.SUM)
            t.copy_((t.float() / self.world).to(t.dtype))

    def _exchange(self, buf: torch.Tensor) -> None:
        if self.local_only:
            return
        d, w = self.dist, self.world
        # AVG is native on RCCL.  With ONE rank (force=True: the path is exercised on a single GPU) the mean is the
        # identity and SUM is asked for instead: RCCL's one-rank AVG runs a separate pre-multiply pass over the whole
        # buffer (oneRankReduce<FuncPreMulSum>, 43 ms per step for the 30 GB arena) that no multi-rank ring contains
        native_avg = buf.is_cuda and self.backend == "nccl" and (w > 1 or self.native_avg_world1)
        if w == 1 and not self.native_avg_world1:
            # (through the staging helpers: a forced one-rank gloo run on a GPU arena must not hand device pointers to a backend
            #  without device collectives — ADVICE r5)
            if self.algo == "allreduce":
                self._all_reduce(buf, d.ReduceOp.SUM)
                self.collectives += 1
            else:
                self._reduce_scatter(buf, buf, d.ReduceOp.SUM)
                self._all_gather(buf, buf)
                self.collectives += 2
            return
        n = buf.numel()
        per = n // w
        if per >= 64:
            per -= per % 16            # every rank's shard starts on a 32-byte boundary of the slice (bf16: 16 elements)
        if self.algo == "allreduce" or per == 0:
            self._avg(buf, native_avg)
            self.collectives += 1
            return
        body = per * w
        shard = buf[self.rank * per:(self.rank + 1) * per]
        if self.reduce_op == "sum" and w > 1:
            self._reduce_scatter(shard, buf[:body], d.ReduceOp.SUM)
        elif native_avg:
            self._reduce_scatter(shard, buf[:body], d.ReduceOp.AVG)
        else:
            self._reduce_scatter(shard, buf[:body], d.ReduceOp.SUM)
            shard.copy_((shard.float() / w).to(shard.dtype))
        self._all_gather(buf[:body], shard)
        self.collectives += 2
        if body < n:
            self._avg(buf[body:], native_avg)
            self.collectives += 1

    def _flush(self) -> None:
        if self._pending_lo is None or (self.world == 1 and not self.force):
            self._pending_lo = self._pending_hi = None
            return
        lo, hi = self._pending_lo, self._pending_hi
        self._pending_lo = self._pending_hi = None
        half = self.comm_dtype == torch.bfloat16
        buf = (self.store.gradc if half else self.store.grad)[lo:hi]
        if not self.local_only:
            self.bytes_reduced += buf.numel() * (2 if half else 4)
        if self.comm_stream is not None:
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            self.store.wait_side(self.comm_stream)
            with torch.cuda.stream(self.comm_stream):
                if self.time_comm and self._t0 is None:
                    self._t0 = torch.cuda.Event(enable_timing=True)
                    self._t0.record(self.comm_stream)
                if half:
                    self._fill_mirror(lo, hi)
                self._exchange(buf)
            if self.after_reduce is not None:
                self.after_reduce(lo, hi, self.comm_stream)
        else:
            if half:
                self._fill_mirror(lo, hi)
            self._exchange(buf)
            if self.after_reduce is not None:
                self.after_reduce(lo, hi, None)

    def reset(self) -> None:
        """forget the slice an aborted backward left pending (single-rank out-of-memory fallback; under data parallelism an
        aborted backward is not recoverable: collectives may already be in flight)"""
        assert self.world == 1, "GradReducer.reset() with peers"
        self._pending_lo = self._pending_hi = None
        self._t0 = None
        if self.plan is not None:
            self._slice_left = [len(sl["buckets"]) for sl in self.plan.slices]
            self._slice_done = [False] * len(self.plan.slices)

    # ---- sharded optimizer step (ShardPlan) ---------------------------------------------------------------------------
    def _exchange_shard(self, buf: torch.Tensor, i: int) -> None:
        """slice i of the plan: reduce-scatter into this rank's shard, the tail all-reduced; no all-gather of gradients"""
        d, sl = self.dist, self.plan.slices[i]
        per, body, n = sl["per"], sl["body"], buf.numel()
        native_avg = buf.is_cuda and self.backend == "nccl" and self.reduce_op == "avg"
        if self.world == 1:
            # one rank (forced; with emulate_world the plan is another world's): the collectives of the real sequence over the
            # whole slice — reduce-scatter in place, the tail's all-reduce
            if per > 0:
                d.reduce_scatter_tensor(buf[:body], buf[:body], op=d.ReduceOp.SUM, group=self.group)
                self.collectives += 1
            if body < n:
                d.all_reduce(buf[body:], op=d.ReduceOp.SUM, group=self.group)
                self.collectives += 1
            return
        if per > 0:
            shard = buf[self.rank * per:(self.rank + 1) * per]
            if self.reduce_op == "sum" or native_avg:
                self._reduce_scatter(shard, buf[:body], d.ReduceOp.SUM if self.reduce_op == "sum" else d.ReduceOp.AVG)
            else:
                self._reduce_scatter(shard, buf[:body], d.ReduceOp.SUM)
                shard.copy_((shard.float() / self.world).to(shard.dtype))
            self.collectives += 1
        if body < n:
            self._avg(buf[body:], native_avg)
            self.collectives += 1

    def _flush_slice(self, i: int) -> None:
        sl = self.plan.slices[i]
        self._slice_done[i] = True
        lo, hi = sl["lo"], sl["hi"]
        half = self.comm_dtype == torch.bfloat16
        buf = (self.store.gradc if half else self.store.grad)[lo:hi]
        self.bytes_reduced += buf.numel() * (2 if half else 4)
        own, tail = self.plan.shard(i), self.plan.tail(i)
        fold = [own] if own[1] > own[0] else []
        if tail[1] > tail[0] and self.plan.rank == 0:
            fold.append(tail)                                  # replicated: every rank holds it, rank 0 counts it
        if self.comm_stream is not None:
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            self.store.wait_side(self.comm_stream)
            with torch.cuda.stream(self.comm_stream):
                if self.time_comm and self._t0 is None:
                    self._t0 = torch.cuda.Event(enable_timing=True)
                    self._t0.record(self.comm_stream)
                if half:
                    self._fill_mirror(lo, hi)
                self._exchange_shard(buf, i)
            if self.after_reduce is not None:
                for a, b in fold:
                    self.after_reduce(a, b, self.comm_stream)
        else:
            if half:
                self._fill_mirror(lo, hi)
            self._exchange_shard(buf, i)
            if self.after_reduce is not None:
                for a, b in fold:
                    self.after_reduce(a, b, None)

    def reduce_scalar(self, t: torch.Tensor) -> torch.Tensor:
        """sum over the ranks of a device scalar, in place (the shards' shares of sum(g^2)) on the current stream"""
        if self.world > 1:
            self._all_reduce(t, self.dist.ReduceOp.SUM)
        elif self.force and not self.local_only:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t

    def _gather_slices(self, arenas_of) -> None:
        """all-gather, slice by slice in forward order on the communication stream, of the shards of the arenas ``arenas_of(slice)``
        names; the buckets of a slice become readable through the store's views once its event has passed (ParamStore._pending)"""
        st, plan = self.store, self.plan
        cur = torch.cuda.current_stream() if self.comm_stream is not None else None
        if self.comm_stream is not None:
            self.comm_stream.wait_stream(cur)                 # the update is enqueued on the compute stream
        ctx = torch.cuda.stream(self.comm_stream) if self.comm_stream is not None else __import__("contextlib").nullcontext()
        with ctx:
            for i, sl in enumerate(plan.slices):
                arenas = arenas_of(sl)
                if sl["per"] == 0 or not arenas:
                    continue                                  # replicated tail only: every rank updated it itself
                lo, body, per = sl["lo"], sl["body"], sl["per"]
                for arena in arenas:
                    whole = arena[lo:lo + body]
                    if self.world == 1:
                        self.dist.all_gather_into_tensor(whole, whole, group=self.group)
                    else:
                        a, b = plan.shard(i)
                        self._all_gather(whole, arena[a:b])
                    self.collectives += 1
                    self.bytes_gathered += whole.numel() * whole.element_size()
                if self.comm_stream is not None:
                    ev = self._gather_events[i]
                    if ev is None:
                        ev = self._gather_events[i] = torch.cuda.Event()
                    ev.record(self.comm_stream)
                    for b_ in sl["buckets"]:
                        st._pending[b_] = ev

    def gather_params(self, overlap: bool = True) -> None:
        """after the sharded update: every rank's freshly written bf16 shadow shard to every rank (what the next forward
        multiplies with), and the fp32 masters of the slices that hold a bucket some forward reads in fp32 (Fp32View: the action
        head) — in fp32 compute mode there is no shadow and the masters travel.  ``overlap``: the next forward waits bucket by
        bucket (the gathers of the later layers run under the first layers' products); False: the compute stream waits here."""
        st = self.store
        fp32_b = st._w32_buckets

        def arenas_of(sl):
            out = [st.shadow] if st.shadow is not None else [st.master]
            if st.shadow is not None and any(b in fp32_b for b in sl["buckets"]):
                out.append(st.master)
            return out
        self._gather_slices(arenas_of)
        if st.shadow is not None:
            st._shadow_version = st.master._version           # the gathers wrote master views: the shadows are NOT stale
        if not overlap:
            st.wait_pending()

    def gather_masters(self) -> None:
        """every rank's fp32 master shard to every rank: afterwards state_dict() / save_pretrained() see the trained weights on
        every rank (the reference's ZeRO-3 save gathers likewise, dexbotic/exp/trainer.py:145-189).  Collective: all ranks call it."""
        st = self.store
        self._gather_slices(lambda sl: [st.master])
        st.wait_pending()
        if st.shadow is not None:
            st._shadow_version = st.master._version

    def bucket_ready(self, b: int) -> None:
        if b in self.skip_buckets:
            return
        lo, hi = self.store.bucket_ranges[b]
        if hi <= lo:
            return
        if self.plan is not None:
            if self.world == 1 and not self.force:
                return
            i = self.plan.slice_of.get(b)
            if i is None or self._slice_done[i]:
                return
            self._slice_left[i] -= 1
            if self._slice_left[i] <= 0:
                self._flush_slice(i)
            return
        if self._pending_lo is None:
            self._pending_lo, self._pending_hi = lo, hi
        elif hi == self._pending_lo or abs(self._pending_lo - hi) < ALIGN:   # backward walks the arena downwards
            self._pending_lo = lo
        elif lo == self._pending_hi or abs(lo - self._pending_hi) < ALIGN:
            self._pending_hi = hi
        else:
            self._flush()
            self._pending_lo, self._pending_hi = lo, hi
        if (self._pending_hi - self._pending_lo) * 4 >= self.min_bucket_bytes:
            self._flush()

    def finish(self) -> None:
        """flush the tail bucket and make the compute stream wait for all collectives"""
        st = self.store
        for b in st.unfired_touched():                     # buckets a frozen/unused slot kept from firing
            st._bucket_fired[b] = True
            self.bucket_ready(b)
        if self.plan is not None and (self.world > 1 or self.force):
            # slices a bucket without any gradient write kept from completing: exchanged as they are if ANY of their buckets was
            # written (the same decision on every rank: same model, same batch structure); untouched slices are left alone
            for i, sl in enumerate(self.plan.slices):
                if not self._slice_done[i] and any(st._bucket_touched[b] for b in sl["buckets"]):
                    self._flush_slice(i)
            self._slice_left = [len(sl["buckets"]) for sl in self.plan.slices]
            self._slice_done = [False] * len(self.plan.slices)
        self._flush()
        if not self.local_only:
            st.invalidate_embed_tracking()                 # other ranks' token rows are now non-zero here too
        if self.comm_stream is not None:
            if self.time_comm and self._t0 is not None:
                t1 = torch.cuda.Event(enable_timing=True)
                t1.record(self.comm_stream)
                self._windows.append((self._t0, t1))
                self._t0 = None
            torch.cuda.current_stream().wait_stream(self.comm_stream)

    def comm_window_ms(self) -> List[float]:
        """per step: time from the first collective's start to the last collective's end on the communication stream
        (``time_comm`` must be on; call after a device synchronize)"""
        out = [a.elapsed_time(b) for a, b in self._windows]
        self._windows = []
        return out
