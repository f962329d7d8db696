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


@dataclass
class Slot:
    name: str
    shape: Tuple[int, ...]
    offset: int
    numel: int
    bucket: int = 0


class ParamStore:
    """Owns the arenas.  ``specs``: ordered list of groups; a group is a list of (name, shape) packed
    contiguously; ``bucket`` ids follow registration order (= forward order) and are the unit of the
    DP all-reduce."""

    def __init__(self, device: torch.device | str, compute_dtype: torch.dtype = torch.float32):
        self.device = torch.device(device)
        self.compute_dtype = compute_dtype
        self.slots: Dict[str, Slot] = {}
        self._groups: List[List[str]] = []
        self._cursor = 0
        self._bucket = 0
        self.bucket_ranges: List[List[int]] = []   # per bucket [lo, hi)
        self.master: Optional[torch.Tensor] = None
        self.shadow: Optional[torch.Tensor] = None
        self.grad: Optional[torch.Tensor] = None
        self.params: Dict[str, nn.Parameter] = {}
        self.grad_written: Dict[str, bool] = {}
        self.on_bucket_ready: Optional[Callable[[int], None]] = None
        self._bucket_pending: List[int] = []
        self._micro_written: set = set()
        self.shadow_t: Optional[torch.Tensor] = None
        self._wt_groups: List[Tuple[Tuple[str, ...], int, int]] = []
        self.wt_index: set = set()
        self._t_stream = None
        self._t_pending = False

    # ---- layout -------------------------------------------------------------------------------------
    def new_bucket(self) -> int:
        self._bucket += 1
        return self._bucket

    def register(self, group: Sequence[Tuple[str, Sequence[int]]]) -> None:
        assert self.master is None, "register() after finalize()"
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
            if self.shadow is not None and self._wt_groups and self.device.type == "cuda":
                self.shadow_t = torch.zeros(total, device=self.device, dtype=self.compute_dtype)
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
    def _view(self, arena: torch.Tensor, names: Sequence[str], shape: Optional[Sequence[int]]) -> torch.Tensor:
        s0 = self.slots[names[0]]
        n = 0
        for nm in names:                      # must be packed back to back
            s = self.slots[nm]
            assert s.offset == s0.offset + n, f"{names} are not adjacent in the arena"
            n += s.numel
        v = arena[s0.offset:s0.offset + n]
        return v.view(tuple(shape)) if shape is not None else v.view(s0.shape) if len(names) == 1 else v

    def w(self, *names: str, shape: Optional[Sequence[int]] = None) -> torch.Tensor:
        """compute-dtype weight view (bf16 shadow, or the fp32 master in fp32 mode); several adjacent
        names give the fused matrix"""
        arena = self.shadow if self.shadow is not None else self.master
        return self._view(arena, names, shape)

    def w32(self, *names: str, shape: Optional[Sequence[int]] = None) -> torch.Tensor:
        return self._view(self.master, names, shape)

    def g(self, *names: str, shape: Optional[Sequence[int]] = None) -> torch.Tensor:
        assert self.grad is not None, "store was finalized with train=False"
        return self._view(self.grad, names, shape)

    # ---- gradient bookkeeping -----------------------------------------------------------------------
    def trainable(self, name: str) -> bool:
        return self.params[name].requires_grad and self.grad is not None

    def accum_flag(self, *names: str) -> bool:
        """True if the gradient slot already holds a value this step (=> kernels must accumulate)."""
        return self.grad_written[names[0]]

    def mark_written(self, *names: str) -> None:
        """``grad_written`` is sticky for the optimizer step (=> later micro-batches accumulate); the bucket
        countdown restarts every micro-batch so the reducer can fire on the last one."""
        for nm in names:
            self.grad_written[nm] = True
            if nm not in self._micro_written:
                self._micro_written.add(nm)
                b = self.slots[nm].bucket
                self._bucket_pending[b] -= 1
                if self._bucket_pending[b] == 0 and self.on_bucket_ready is not None and not self._bucket_fired[b]:
                    self._bucket_fired[b] = True
                    self.on_bucket_ready(b)

    def set_expected(self, exclude: Iterable[str] = ()) -> None:
        """(re)count, per bucket, the slots a backward pass is expected to write: trainable and not in
        ``exclude`` (parameters the path never touches, e.g. lm_head / the unused last CLIP layer)."""
        exclude = set(exclude)
        self._bucket_total = [0] * len(self.bucket_ranges)
        for s in self.slots.values():
            if self.params[s.name].requires_grad and s.name not in exclude:
                self._bucket_total[s.bucket] += 1
        self.begin_micro()

    def begin_micro(self) -> None:
        self._micro_written = set()
        self._bucket_pending = list(self._bucket_total)
        self._bucket_fired = [False] * len(self._bucket_total)

    def begin_step(self, zero_names: Iterable[str] = ()) -> None:
        """Reset per-step state.  Gradients are produced with beta=0 writes by the GEMM kernels, so
        only slots that are accumulated into (embedding rows, never-written slots) need zeroing."""
        for nm in self.grad_written:
            self.grad_written[nm] = False
        self.begin_micro()
        for nm in zero_names:
            self.g(nm).zero_()

    def never_written(self) -> List[str]:
        return [nm for nm, wtn in self.grad_written.items() if not wtn and self.params[nm].requires_grad]

    def attach_grads(self) -> None:
        """expose the arena views as ``param.grad`` (HF Trainer / user code compatibility)"""
        for nm, p in self.params.items():
            if p.requires_grad and self.grad is not None:
                p.grad = self.g(nm)

    def sync_shadow(self) -> None:
        """re-derive the bf16 shadows from the fp32 masters (after load_state_dict / init)"""
        if self.shadow is not None:
            from . import kernels as K
            K.cast(self.master, self.shadow.dtype, out=self.shadow)
        self.sync_transposed()

    # ---- transposed bf16 weight shadows: dX = dY W runs as an NT product on the fast MFMA path -------------
    def register_wt(self, names: Sequence[str], rows: int, cols: int) -> None:
        """the (fused) weight `names` = [rows, cols] also needs a [cols, rows] copy for its input gradient"""
        self._wt_groups.append((tuple(names), int(rows), int(cols)))
        self.wt_index.add(tuple(names))

    @property
    def has_wt(self) -> bool:
        return self.shadow_t is not None

    def wt(self, *names: str, shape: Sequence[int]) -> torch.Tensor:
        return self._view(self.shadow_t, names, shape)

    def sync_transposed(self, overlap: bool = False) -> None:
        """refresh the W^T shadows.  With ``overlap`` the ~200 HBM-bound transposes run on a side HIP stream
        (they are first needed by the NEXT backward, so they hide under the next forward's MFMA work);
        consumers call wait_transposed()."""
        if self.shadow_t is None:
            return
        from . import kernels as K
        if overlap and self.device.type == "cuda":
            if self._t_stream is None:
                self._t_stream = torch.cuda.Stream(device=self.device)
            self._t_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._t_stream):
                for names, rows, cols in self._wt_groups:
                    K.transpose(self.w(*names, shape=(rows, cols)), out=self.wt(*names, shape=(cols, rows)))
            self._t_pending = True
            return
        for names, rows, cols in self._wt_groups:
            K.transpose(self.w(*names, shape=(rows, cols)), out=self.wt(*names, shape=(cols, rows)))

    def wait_transposed(self) -> None:
        if self._t_pending:
            torch.cuda.current_stream().wait_stream(self._t_stream)
            self._t_pending = False


class Fp32View:
    """Same interface as ParamStore but ``w()`` hands out the fp32 masters: used by the diffusion action
    head, which the reference keeps in fp32 (cogact_arch.py:133, SURVEY.md App. A dtype notes)."""

    has_wt = False        # fp32 GEMMs use the exact NN / TN kernels

    def __init__(self, store: ParamStore):
        self._s = store

    def w(self, *names: str, shape=None):
        return self._s.w32(*names, shape=shape)

    def __getattr__(self, item):
        return getattr(self._s, item)


def attach_parameters(root: nn.Module, store: ParamStore, containers: Optional[Dict[str, nn.Module]] = None) -> None:
    """Register every arena view as an nn.Parameter under its dotted name so that ``root.state_dict()``
    has exactly the reference's keys.  Intermediate modules that do not exist yet
    are created as empty containers; existing modules (e.g. the backbone / tower objects) are reused."""
    for name, p in store.params.items():
        parts = name.split(".")
        mod = root
        for part in parts[:-1]:
            child = mod._modules.get(part)
            if child is None:
                child = nn.Module()
                mod.add_module(part, child)
            mod = child
        mod.register_parameter(parts[-1], p)


# ------------------------------------------------------------------------------------------ optimizer
def no_decay_name(name: str) -> bool:
    """Parameter-group rule of OptimizerConfig._get_optimizer_grouped_parameters (base_exp.py:95-203)
    under the reference's pinned transformers: no weight decay for parameters of nn.LayerNorm modules
    and for every name containing "bias" (RMSNorm weights DO decay there)."""
    if "bias" in name:
        return True
    return any(t in name for t in ("layer_norm1.", "layer_norm2.", "pre_layrnorm.", "post_layernorm."))


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
                 exclude: Iterable[str] = ()):
        """``exclude``: parameters that never receive a gradient (torch.optim skips ``grad is None``
        parameters entirely — no weight decay either)."""
        from . import kernels as K  # noqa: F401  (fail early if the library is missing)
        self.store, self.cfg = store, cfg
        exclude = set(exclude)
        dev = store.device
        self.m = torch.zeros_like(store.master)
        self.v = torch.zeros_like(store.master)
        self.step_count = 0
        prefixes = prefixes or {"mm_projector": "mm_projector", "mm_vision": "mm_vision", "action_head": "action_head"}
        # groups: (lr_key, decay?) -> index  (<= 8 groups, as the reference builds them)
        self.group_keys: List[Tuple[str, bool]] = []
        cs, cl, cg = [], [], []
        for s in sorted(store.slots.values(), key=lambda s: s.offset):
            if not store.params[s.name].requires_grad or s.name in exclude:
                continue
            lr_key = "base"
            if cfg.mm_projector_lr is not None and prefixes["mm_projector"] in s.name:
                lr_key = "mm_projector"
            elif cfg.mm_vision_lr is not None and prefixes["mm_vision"] in s.name:
                lr_key = "mm_vision"
            elif cfg.action_head_lr is not None and prefixes["action_head"] in s.name:
                lr_key = "action_head"
            key = (lr_key, not no_decay_name(s.name))
            if key not in self.group_keys:
                self.group_keys.append(key)
            gi = self.group_keys.index(key)
            o = 0
            while o < s.numel:
                ln = min(cfg.chunk, s.numel - o)
                cs.append(s.offset + o)
                cl.append(ln)
                cg.append(gi)
                o += ln
        assert len(self.group_keys) <= 8
        self.chunk_start = torch.tensor(cs, dtype=torch.int64, device=dev)
        self.chunk_len = torch.tensor(cl, dtype=torch.int32, device=dev)
        self.chunk_grp = torch.tensor(cg, dtype=torch.int32, device=dev)
        self.sumsq = torch.zeros(1, device=dev, dtype=torch.float32)
        self.norm = torch.zeros(1, device=dev, dtype=torch.float32)
        self.coef = torch.ones(1, device=dev, dtype=torch.float32)
        self.scratch = torch.empty(4096, device=dev, dtype=torch.float64)

    def _lrs_wds(self, lr_scale: float):
        c = self.cfg
        lr_of = {"base": c.base_lr, "mm_projector": c.mm_projector_lr, "mm_vision": c.mm_vision_lr,
                 "action_head": c.action_head_lr}
        lrs = [lr_of[k] * lr_scale for k, _ in self.group_keys]
        wds = [c.weight_decay if dec else 0.0 for _, dec in self.group_keys]
        return lrs, wds

    def step(self, lr_scale: float = 1.0, sumsq: Optional[torch.Tensor] = None) -> None:
        """``sumsq``: device scalar holding sum(g^2) over the arena if someone already accumulated it (GradNormTracker
        does, bucket by bucket under the backward); otherwise one pass over the gradient arena computes it here."""
        from . import kernels as K
        st, c = self.store, self.cfg
        self.step_count += 1
        clip = None
        if c.max_grad_norm is not None:
            if sumsq is None:
                K.sumsq(st.grad, self.sumsq, self.scratch)
                sumsq = self.sumsq
            K.clip_coef(sumsq, float(c.max_grad_norm), self.norm, self.coef)
            clip = self.coef
        lrs, wds = self._lrs_wds(lr_scale)
        K.adamw(st.master, st.grad, self.m, self.v, st.shadow, self.chunk_start, self.chunk_len, self.chunk_grp,
                lrs, wds, c.adam_beta1, c.adam_beta2, c.adam_epsilon, self.step_count, clip=clip)
        st.sync_transposed(overlap=True)


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
        dev = store.device
        self.acc = torch.zeros(1, device=dev, dtype=torch.float32)
        self.scratch = torch.empty(4096, device=dev, dtype=torch.float64)
        self.stream = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None
        self._lo: Optional[int] = None
        self._hi: Optional[int] = None

    def begin(self) -> None:
        self.acc.zero_()                       # on the compute stream; every fold waits for that stream first
        self._lo = self._hi = None

    def fold(self, lo: int, hi: int, after=None) -> None:
        """acc += sum(grad[lo:hi]^2) on the tracker's side stream, ordered after everything enqueued so far on ``after``
        (default: the compute stream; the communication stream when the slice has just been all-reduced there — the
        collectives that follow on that stream are not held up by the reduction)"""
        from . import kernels as K
        if hi <= lo:
            return
        if self.stream is None:
            K.sumsq(self.store.grad[lo:hi], self.acc, self.scratch, accumulate=True)
            return
        self.stream.wait_stream(after if after is not None else torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            K.sumsq(self.store.grad[lo:hi], self.acc, self.scratch, accumulate=True)

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
            for b in reversed(range(len(st.bucket_ranges))):
                if not st._bucket_fired[b] and st._bucket_pending[b] < st._bucket_total[b]:
                    st._bucket_fired[b] = True
                    self.bucket_ready(b)
        if self._lo is not None:
            self.fold(self._lo, self._hi)
            self._lo = self._hi = None
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)
        return self.acc


# ---------------------------------------------------------------------------------------- DP reducer
class GradReducer:
    """Data-parallel gradient averaging over RCCL (torch.distributed backend "nccl" on ROCm; "gloo" in
    the CPU tests).  Buckets are contiguous slices of the gradient arena in reverse forward order
    (DiT head -> LLM layers 27..0 -> projector -> ViT); a bucket is all-reduced in place on a side HIP
    stream as soon as every slot in it has been written by the backward, so communication of block i
    overlaps the backward compute of blocks < i.  xGMI is point-to-point (7 links x ~153 GB/s):
    buckets are merged up to ``min_bucket_bytes`` so each collective is large enough to drive all
    links, and parameters that never receive a gradient (lm_head, the unused last CLIP layer) are
    never sent (the reference papers over them with ddp_find_unused_parameters=True, trainer.py:121).
    """

    def __init__(self, store: ParamStore, group=None, min_bucket_bytes: int = 256 << 20,
                 skip: Iterable[str] = (), force: bool = False, comm_dtype: torch.dtype = torch.float32):
        import torch.distributed as dist
        self.dist = dist
        self.force = force          # run the collectives even at world size 1 (exercises the RCCL path)
        # fp32: the arena slice is averaged in place.  bf16: the slice is cast to a bf16 staging buffer on the
        # communication stream, averaged, and cast back — half the xGMI bytes; this is what the reference's
        # DeepSpeed ZeRO-2 bf16 run reduces (its gradients are bf16 tensors, script/deepspeed/zero2.json).
        assert comm_dtype in (torch.float32, torch.bfloat16)
        self.comm_dtype = comm_dtype
        self.store = store
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.min_bucket_bytes = min_bucket_bytes
        self.skip_buckets = set()
        self.comm_stream = torch.cuda.Stream(device=store.device) if store.device.type == "cuda" else None
        self._pending_lo: Optional[int] = None
        self._pending_hi: Optional[int] = None
        self._handles: List = []
        self.after_reduce = None    # callable(lo, hi, comm_stream): called once a slice's collective (and cast back) is enqueued
        self.bytes_reduced = 0
        skip = set(skip)
        # buckets whose every slot is skipped (never gets a gradient) are not communicated
        by_bucket: Dict[int, List[str]] = {}
        for s in store.slots.values():
            by_bucket.setdefault(s.bucket, []).append(s.name)
        for b, names in by_bucket.items():
            if all((n in skip) or (not store.params[n].requires_grad) for n in names):
                self.skip_buckets.add(b)
        store.on_bucket_ready = self.bucket_ready

    def _flush(self) -> None:
        if self._pending_lo is None or (self.world == 1 and not self.force):
            self._pending_lo = self._pending_hi = None
            return
        lo, hi = self._pending_lo, self._pending_hi
        self._pending_lo = self._pending_hi = None
        buf = self.store.grad[lo:hi]
        half = self.comm_dtype == torch.bfloat16
        self.bytes_reduced += buf.numel() * (2 if half else 4)
        if self.comm_stream is not None:
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                if half:
                    from . import kernels as K
                    stage = K.cast(buf, torch.bfloat16)      # allocated and freed on the communication stream
                    self.dist.all_reduce(stage, op=self.dist.ReduceOp.AVG, group=self.group)
                    K.cast(stage, torch.float32, out=buf)
                else:
                    self.dist.all_reduce(buf, op=self.dist.ReduceOp.AVG, group=self.group)
            if self.after_reduce is not None:
                self.after_reduce(lo, hi, self.comm_stream)
        else:  # CPU / gloo: no AVG op
            if half:
                stage = buf.to(torch.bfloat16)
                self.dist.all_reduce(stage, op=self.dist.ReduceOp.SUM, group=self.group)
                buf.copy_(stage.float() / self.world)
            else:
                self.dist.all_reduce(buf, op=self.dist.ReduceOp.SUM, group=self.group)
                buf.div_(self.world)
            if self.after_reduce is not None:
                self.after_reduce(lo, hi, None)

    def bucket_ready(self, b: int) -> None:
        if b in self.skip_buckets:
            return
        lo, hi = self.store.bucket_ranges[b]
        if hi <= lo:
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
        for b in reversed(range(len(st.bucket_ranges))):   # buckets a frozen/unused slot kept from firing
            if not st._bucket_fired[b] and st._bucket_pending[b] < st._bucket_total[b]:
                st._bucket_fired[b] = True
                self.bucket_ready(b)
        self._flush()
        if self.comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
