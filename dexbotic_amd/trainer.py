"""Minimal fine-tune driver for the native policy: what HF ``Trainer.training_step`` + ``optimizer.step`` do
for the reference (exp/trainer.py:25-36,88-138; exp/base_exp.py:865-871), on the flat arenas.

    step(batch):  begin_step -> model(**batch).loss.backward() -> [DP all-reduce finishes] ->
                  global-norm clip + fused AdamW (+ bf16 shadow refresh) -> cosine lr

One process per GPU; with ``torch.distributed`` initialised (backend "nccl" = RCCL over xGMI) the gradient
arena is averaged across ranks by engine.GradReducer, overlapped with the backward.
"""
from __future__ import annotations

import os

from typing import Dict, List, Optional

import torch

from . import kernels as K

from .engine import FusedAdamW, GradNormTracker, GradReducer, OptimConfig, cosine_lr_scale


# ---------------------------------------------------------------------------------- micro-batch coalescing
_ROW_KEYS = ("input_ids", "attention_mask", "labels", "images", "actions", "states")     # leading dimension B
_DRAW_KEYS = ("noise", "timesteps", "drop_ids")                                          # leading dimension R * B, index r * B + b
_PAD = {"input_ids": 0, "attention_mask": 0, "labels": -100}                             # what a right-padded position holds


def coalesce_batches(batches: List[Dict[str, torch.Tensor]]) -> Optional[Dict[str, torch.Tensor]]:
    """The micro-batches of ONE optimizer step as a single batch, or None when they cannot be put together (keys this function
    does not know, unequal micro-batch sizes — the mean over the merged batch would weigh the samples differently — or
    unequal trailing shapes).  Token rows of unequal length are right-padded to the longest (pad id 0 / mask 0 / label -100:
    the splice drops masked positions before it lays the sequence out, dexbotic_arch.py:182-373, so the result depends on the
    real tokens only).  Injected draws ([R * B, ...], row r * B + b, cogact_arch.py:110-125) are re-interleaved per repeat."""
    keys = set(batches[0])
    if any(set(b) != keys for b in batches) or not keys <= set(_ROW_KEYS + _DRAW_KEYS):
        return None
    if not all(torch.is_tensor(b[k]) for b in batches for k in keys) or "input_ids" not in keys:
        return None
    B = batches[0]["input_ids"].shape[0]
    if any(b["input_ids"].shape[0] != B for b in batches):
        return None
    out = {}
    for k in keys & set(_ROW_KEYS):
        vs = [b[k] for b in batches]
        if any(v.shape[0] != B or v.dim() != vs[0].dim() or v.dtype != vs[0].dtype or v.device != vs[0].device for v in vs):
            return None
        if k in _PAD and vs[0].dim() == 2:
            S = max(v.shape[1] for v in vs)
            if "attention_mask" not in keys and any(v.shape[1] != S for v in vs):
                # no mask (= every position valid, splice.py): padding a shorter micro-batch would turn pad ids into real
                # tokens of its samples — such a group runs pass by pass
                return None
            vs = [v if v.shape[1] == S else torch.nn.functional.pad(v, (0, S - v.shape[1]), value=_PAD[k]) for v in vs]
        if any(v.shape[1:] != vs[0].shape[1:] for v in vs):
            return None
        out[k] = torch.cat(vs, dim=0)
    for k in keys & set(_DRAW_KEYS):
        vs = [b[k] for b in batches]
        if any(v.shape != vs[0].shape or v.shape[0] % B != 0 or v.dtype != vs[0].dtype or v.device != vs[0].device for v in vs):
            return None
        R = vs[0].shape[0] // B
        out[k] = torch.cat([v.reshape(R, B, *v.shape[1:]) for v in vs], dim=1).reshape(R * B * len(vs), *vs[0].shape[1:])
    return out


class NativeTrainer:
    def __init__(self, model, optim: Optional[OptimConfig] = None, total_steps: int = 0, warmup_steps: int = 0,
                 grad_accum: int = 1, distributed: Optional[bool] = None, min_bucket_bytes: int = 256 << 20,
                 force_reducer: bool = False, grad_comm_dtype: torch.dtype = torch.float32, grad_sync: str = "rs_ag",
                 grad_dtype: torch.dtype = torch.float32, overlap_optimizer: bool = False, optimizer_groups=None,
                 native_avg_world1: bool = False, coalesce_micro_batches: Optional[bool] = None, grad_reduce_op: str = "sum",
                 shard_optimizer: Optional[bool] = None, emulate_world: int = 0, gather_overlap: bool = True):
        import torch.distributed as dist
        from . import hostcpu
        # torch's intra-op pool inside the cgroup's CPU quota (hostcpu.py: a pool sized by the host's 256 CPUs under a 16-CPU quota
        # gets the launching thread throttled in the middle of a step); DXA_HOST_THREADS=0 leaves torch alone
        self.host_threads = hostcpu.limit_host_threads()
        self.model = model
        # shard_optimizer: the sharded optimizer step under data parallelism (engine.ShardPlan — the reference's default DeepSpeed
        # ZeRO config partitions optimizer state and update, base_exp.py:229 / zero3.json): reduce-scatter of the gradients, sum(g^2)
        # over the own shard + one scalar all-reduce, adamw_k over 1 / world of the arena, all-gather of the updated bf16 shadows
        # (``gather_overlap``: under the next forward, bucket by bucket).  Default: on for world size > 1 with grad_sync="rs_ag"
        # (env DXA_SHARD_OPT=0: every rank repeats the full update on all-gathered gradients, rounds 1-5).  Parameters are
        # bit-identical to the replicated path's (tests/test_zz_dp2_gpu.py).  ``emulate_world``: engine.GradReducer.
        if shard_optimizer is None:
            multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
            shard_optimizer = os.environ.get("DXA_SHARD_OPT", "1") != "0" and grad_sync == "rs_ag" and not overlap_optimizer \
                and (multi or emulate_world > 1)
        self.gather_overlap = bool(gather_overlap) and os.environ.get("DXA_GATHER_OVERLAP", "1") != "0"
        # coalesce_micro_batches: the ``grad_accum`` micro-batches of an optimizer step run as ONE forward / backward over their
        # concatenation (coalesce_batches).  Gradient accumulation exists in the reference recipe (8 episodes x 2, cogact_exp.py:
        # 41-46) to fit 80 GB parts; the mean loss over the merged batch IS the mean of the micro-batch means, so the step is the
        # same up to fp32 summation order, and on 288 GB it runs at the one-pass rate (full 256-row tile grids, one dW product,
        # half the small launches).  Only for models that declare ``coalescible_micro_batches`` (stateless in the batch: not
        # MemVLA, whose bank walks the batch in order).  Default: env DEXBOTIC_AMD_COALESCE (off); exp/trainer.NativeDexboticTrainer
        # turns it on.  A group that cannot be merged — or that runs out of memory merged — runs micro-batch by micro-batch.
        if coalesce_micro_batches is None:
            coalesce_micro_batches = os.environ.get("DEXBOTIC_AMD_COALESCE", "0") != "0"
        self.coalesce = bool(coalesce_micro_batches) and bool(getattr(model, "coalescible_micro_batches", False))
        self._held: list = []
        self.coalesced_steps = 0
        self.store = model.store
        self.cfg = optim or OptimConfig()
        unused = list(model.unused_parameter_names()) if hasattr(model, "unused_parameter_names") else []
        # overlap_optimizer: the fused AdamW runs bucket by bucket in forward order on a side stream and the NEXT step's
        # forward waits per bucket (engine.FusedAdamW(overlap=True)).  step() then returns with the update still in flight:
        # everything that goes through the store's views (the model's own forward / inference, sync_shadow) orders itself;
        # code that reads parameters directly (p.data, state_dict()) calls trainer.synchronize() first.
        # optimizer_groups: explicit parameter groups [{"names": [...]}, ...] (exp/trainer.NativeDexboticTrainer: the groups the
        # exp's OptimizerConfig built); default: engine.FusedAdamW's own name rule.
        self.total_steps, self.warmup_steps, self.grad_accum = total_steps, warmup_steps, grad_accum
        self.global_step = 0
        self.micro = 0
        self.store.set_expected(unused)
        use_dist = (dist.is_available() and dist.is_initialized()) if distributed is None else distributed
        # grad_dtype = bfloat16: the reference's DeepSpeed bf16 recipe (bf16 gradients, fp32 masters in the optimizer): the dW
        # products write the bf16 gradient arena only, AdamW and the norm read it.  Needs one micro-batch per step (accumulating
        # micro-batches in bf16 would lose bits) and a bf16 compute model; implies bf16 exchange under data parallelism.
        bf16_grads = grad_dtype == torch.bfloat16 and grad_accum == 1 and self.store.compute_dtype == torch.bfloat16 \
            and self.store.device.type == "cuda"
        if bf16_grads:
            grad_comm_dtype = torch.bfloat16
        self.reducer = None
        if use_dist and (dist.get_world_size() > 1 or force_reducer):
            # native_avg_world1: at world size 1 take the exact collective sequence of N > 1 (bench.py --native-avg: RCCL kernels
            # sharing the GPU with the backward's GEMM grids — the contention measurement of DESIGN.md section 6)
            self.reducer = GradReducer(self.store, min_bucket_bytes=min_bucket_bytes, skip=unused, force=force_reducer,
                                       comm_dtype=grad_comm_dtype, algo=grad_sync, native_avg_world1=native_avg_world1,
                                       reduce_op=grad_reduce_op, shard=bool(shard_optimizer), emulate_world=emulate_world)
        elif bf16_grads:
            self.reducer = GradReducer(self.store, min_bucket_bytes=min_bucket_bytes, skip=unused, comm_dtype=torch.bfloat16,
                                       local_only=True)
        self.store.bf16_grads = bf16_grads
        self.sharded = self.reducer is not None and self.reducer.plan is not None
        self.opt = FusedAdamW(self.store, self.cfg, exclude=unused, overlap=overlap_optimizer and not self.sharded,
                              groups=optimizer_groups, ranges=self.reducer.plan.owned() if self.sharded else None)
        # global-norm clip: sum(g^2) is folded in bucket by bucket under the backward (after the all-reduce under DP)
        self.norm_tracker = None
        if self.cfg.max_grad_norm is not None and self.store.device.type == "cuda":
            self.norm_tracker = GradNormTracker(self.store, min_bytes=min_bucket_bytes)
            if self.reducer is not None:
                self.reducer.after_reduce = lambda lo, hi, stream: self.norm_tracker.fold(lo, hi, stream)
                self.norm_tracker.src = self.reducer.result_arena
        self.store.attach_grads()
        # single GPU: nobody but the splice backward writes the dense embedding gradient, so it can be re-zeroed row-wise
        local = self.reducer is None or self.reducer.local_only        # nobody else writes this rank's gradient arenas
        self.store.sparse_embed_zero = local
        # single GPU: sum(g^2) of the weight gradients comes out of the dW products' epilogues (under gradient accumulation:
        # of the LAST micro-batch's, which store the step's final values — micro_step keeps store.last_micro current)
        self.store.epi_sumsq = local and self.norm_tracker is not None
        self.store.defer_wgrad = True           # parameters with several consumers per forward: one dW product for all of them
        self.store.invalidate_embed_tracking()
        self._zeroed_unused = False
        self.store.managed = True               # this object calls begin_step / begin_micro (the model's pre-hook stands down)
        # Gradient work that nothing downstream in the backward chain waits for runs on a side HIP stream, beside the dX chain
        # (ParamStore.wgrad_stream; the compute stream joins before a bucket's completion hook, the norm / reducer finish and the
        # optimizer).  DXA_WGRAD_STREAM = 3 (default): the fp32 action head's dW products (1088 rows: a fifth of the CUs each) and
        # the bias gradients' column sums — 244.7 -> 241.3 ms per step, bit-identical gradients, the big products' launch times
        # unchanged (profiles/r04_wgrad_stream_abc.txt); 2: the head's dW only (242.7); 1: EVERY dW product (240.1 on another box,
        # but a 16-bit dW beside the next dX shares the CUs with it: each launch then takes 400-500 us and the per-launch roofline
        # of section 5 stops meaning anything); 0: everything on the compute stream.  Default per model (``gradient_side_stream``):
        # on for DB-CogACT and — since round 6, when its 7,200 launches stopped being host-bound: 310.7 -> 307.8 ms — for MemVLA
        # (round 4, 9,000 host-bound launches: 363 -> 372 ms with it, a cross-stream dependency per product) and pi0 (250.0 -> 245.3 ms; round 4, before its host-side waits were removed: 271.3 vs 270.1, profiles/r04_wgrad_stream_pi0.txt).
        mode = os.environ.get("DXA_WGRAD_STREAM", "3" if getattr(model, "gradient_side_stream", False) else "0")
        if mode != "0" and self.store.device.type == "cuda":
            self.store.wgrad_stream = torch.cuda.Stream(device=self.store.device)
            self.store.wgrad_stream_f32_only = mode in ("2", "3")
            self.store.bgrad_on_side = mode == "3"
        self.update_due = False
        self._sumsq, self._reducing = None, False
        self.last_output = None
        self._gathered_moments = None

    def set_grad_accum(self, n: int) -> None:
        """change the number of micro-batches per optimizer step (between optimizer steps only)"""
        assert not self.update_due and self.micro % self.grad_accum == 0 and not self._held, \
            "set_grad_accum() in the middle of an optimizer step"
        assert n == 1 or not self.store.bf16_grads, "bf16 gradient arena needs one micro-batch per step"
        self.grad_accum, self.micro = int(n), 0
        local = self.reducer is None or self.reducer.local_only
        self.store.epi_sumsq = local and self.norm_tracker is not None

    def close_short_group(self) -> None:
        """An optimizer step is wanted although the accumulation group is not full: HF closes a SHORT group at the end of an
        epoch whose length does not divide by gradient_accumulation_steps (Trainer._inner_training_loop) and calls
        ``optimizer.step()``.  transformers 5.x announces the group's size (``current_gradient_accumulation_steps``, followed by
        exp/trainer.NativeDexboticTrainer.training_step); 4.51 — the version the reference pins — does not, so the optimizer
        facade lands here.  What was handed over so far IS the group; every micro-batch keeps the scale it was given (default
        1 / grad_accum of the NOMINAL size, as HF 4.51 scales it), and the data-parallel exchange / the sum of squares that the
        last micro-batch of a full group brings to completion are brought to completion now.  The losses of held micro-batches
        were already reported as 0 to the caller and stay unreported."""
        assert not self.update_due
        nominal = self.grad_accum
        if self._held:                                    # coalescing: nothing has run yet
            held, self._held = self._held, []
            n = len(held)
            scales = [float(sc) if sc is not None else 1.0 / nominal for _, sc in held]
            self.grad_accum, self.micro = n, 0
            try:
                merged = coalesce_batches([b for b, _ in held]) if len(set(scales)) == 1 else None
                if merged is not None:
                    self._micro(merged, scales[0] * n, group=n)
                else:
                    for (b, _), sc in zip(held, scales):
                        self._micro(b, sc)
            finally:
                self.grad_accum, self.micro = nominal, 0
            return
        k = self.micro % nominal
        assert k > 0, "close_short_group() with nothing accumulated"
        st = self.store
        if st._accum_stash:
            from .functional import flush_accum
            st.last_micro = True
            flush_accum(st)                               # (dY, X) pairs held for a second micro-batch that never came
        st.join_wgrad()
        if not self._zeroed_unused:
            for nm in st.never_written():
                st.g(nm).zero_()
            self._zeroed_unused = True
        reducing = self.reducer is not None and (self.reducer.world > 1 or self.reducer.force)
        self._sumsq = None                                # FusedAdamW.step takes the norm in one pass over the arena
        if reducing:
            st._mirrored = set()                          # bf16 exchange: cast every written slot (no epilogue mirrored the sum)
            st._bucket_fired = [True] * len(st.bucket_ranges)
            st._bucket_touched = [True] * len(st.bucket_ranges)
            # the reducer's after_reduce folds every exchanged slice into the norm tracker: bracketed by begin() / finish() like in
            # a full group (round 5 left those folds on the tracker's stream with nothing joining it — ADVICE r5), and their sum
            # IS the step's sum of squares (the sharded step has no other: its shards' shares are all-reduced below)
            if self.norm_tracker is not None:
                self.norm_tracker.begin()
            for b in reversed(range(len(st.bucket_ranges))):
                self.reducer.bucket_ready(b)
            self.reducer.finish()
            if self.norm_tracker is not None:
                self._sumsq = self.norm_tracker.finish(fire_unfired=False)
                if self.sharded:
                    self.reducer.reduce_scalar(self._sumsq)
        self._reducing = reducing
        self.update_due = True
        self.micro = 0

    def synchronize(self) -> None:
        """make the current stream wait for an overlapped optimizer update still in flight"""
        self.store.wait_pending()

    def lr_scale(self) -> float:
        if self.total_steps <= 0:
            return 1.0
        return cosine_lr_scale(self.global_step, self.total_steps, self.warmup_steps)

    def step(self, batch: Dict[str, torch.Tensor]) -> torch.Tensor:
        """one micro-batch; the optimizer runs every ``grad_accum`` calls.  Returns the (detached) loss."""
        loss = self.micro_step(batch)
        if self.update_due:
            self.apply_update()
        return loss

    def micro_step(self, batch: Dict[str, torch.Tensor], loss_scale: Optional[float] = None) -> torch.Tensor:
        """forward + backward of one micro-batch into the gradient arenas; on the last micro-batch of an optimizer step the
        data-parallel exchange and the sum of squares are brought to completion and ``update_due`` is set.  Split from
        ``apply_update`` so that a loop which owns the optimizer call — HF ``Trainer``: ``training_step`` per micro-batch,
        ``optimizer.step()`` at the boundary (exp/trainer.NativeDexboticTrainer) — drives the same machinery.
        ``loss_scale``: factor on the loss before backward; default 1 / grad_accum (the mean over micro-batches).
        With ``coalesce`` the first grad_accum - 1 calls of a group only hold their batch (and return a zero loss); the last
        call runs the whole group in one pass and returns the SUM of the group's micro-batch losses, so that a caller that
        adds up what the calls return (HF ``Trainer``: tr_loss) sees the same total."""
        if not (self.coalesce and self.grad_accum > 1):
            return self._micro(batch, loss_scale)
        self._held.append((batch, loss_scale))
        if len(self._held) < self.grad_accum:
            return torch.zeros((), device=self.store.device, dtype=torch.float32)
        held, self._held = self._held, []
        n = len(held)
        scales = {sc for _, sc in held}
        merged = coalesce_batches([b for b, _ in held]) if len(scales) == 1 else None
        if merged is not None:
            sc = held[0][1]
            if self.reducer is not None and self.reducer.world > 1:
                # data parallel: no out-of-memory fallback — the aborted backward of ONE rank may already have fired bucket
                # collectives its peers are waiting in, and the ranks would disagree on ``coalesce`` afterwards
                loss = self._micro(merged, 1.0 if sc is None else float(sc) * n, group=n)
                self.coalesced_steps += 1
                return loss * n
            try:
                loss = self._micro(merged, 1.0 if sc is None else float(sc) * n, group=n)
                self.coalesced_steps += 1
                return loss * n
            except torch.OutOfMemoryError:
                # the merged pass does not fit: this and every later group runs micro-batch by micro-batch (begin_step of the
                # first one starts the gradient arenas afresh: every slot's first write replaces).  The pending dW products and
                # accumulation pairs of the aborted backward are DROPPED, not run: their operands belong to a graph that is
                # being torn down, and running them would allocate while the failed pass still holds its memory
                self.coalesce = False
                self.store.drop_pending_wgrads()
                self.last_output = None
                if self.reducer is not None:
                    self.reducer.reset()
                if self.norm_tracker is not None:
                    self.norm_tracker.begin()
                torch.cuda.empty_cache()
        total = None
        for b, sc in held:
            l_ = self._micro(b, sc)
            total = l_ if total is None else total + l_
        return total

    def _micro(self, batch: Dict[str, torch.Tensor], loss_scale: Optional[float] = None, group: Optional[int] = None) -> torch.Tensor:
        """one pass; ``group`` = n: the pass covers a whole accumulation group of n micro-batches (first and last at once)"""
        first = group is not None or self.micro % self.grad_accum == 0
        last = group is not None or (self.micro + 1) % self.grad_accum == 0
        self.store.last_micro = last
        # exactly two micro-batches: dW of the linears = one product over both (the pair of micro-batch 1 is held until 2)
        self.store.accum_merge = group is None and self.grad_accum == 2 and self.store.device.type == "cuda" \
            and not self.store.bf16_grads and os.environ.get("DXA_NO_ACCUM_MERGE") is None
        if first:
            self.store.begin_step()
        else:
            self.store.begin_micro()
        # with accumulation, communication happens on the last micro-batch only (like DDP.no_sync)
        hook = None
        if last:
            if self.reducer is not None and (self.reducer.world > 1 or self.reducer.force):
                hook = self.reducer.bucket_ready
            elif self.norm_tracker is not None:
                hook = self.norm_tracker.bucket_ready
            if self.norm_tracker is not None:
                self.norm_tracker.begin()
        self.store.on_bucket_ready = hook
        self.store._record_w32 = True           # (which buckets are read as fp32 masters: the sharded step gathers those in fp32)
        try:
            with K.f32_gemm_mode(getattr(self.model.config, "fp32_matmul", "exact")):
                out = self.model(**batch)
                loss = out.loss
                self.store.wait_pending()           # (overlapped optimizer: buckets the forward never touched)
                scale = (1.0 / self.grad_accum) if loss_scale is None else float(loss_scale)
                (loss * scale if scale != 1.0 else loss).backward()
        finally:
            self.store._record_w32 = False
        self.last_output = out
        self.store.flush_wgrads()               # (only when autograd pruned a consumer of a multiply-used parameter)
        if last and self.store._accum_stash:
            from .functional import flush_accum
            flush_accum(self.store)             # (a parameter the last micro-batch did not use)
        self.store.join_wgrad()                 # (dW products on the side stream, if any)
        self.micro += 1 if group is None else group
        if last:
            if not self._zeroed_unused:
                # slots no kernel ever writes (lm_head, unused CLIP layer, history_embedder) stay exactly zero
                for nm in self.store.never_written():
                    self.store.g(nm).zero_()
                self._zeroed_unused = True
            reducing = self.reducer is not None and (self.reducer.world > 1 or self.reducer.force)
            if self.reducer is not None:
                self.reducer.finish()
            self._sumsq = None
            if self.norm_tracker is not None:
                self._sumsq = self.norm_tracker.finish(fire_unfired=not reducing)
                if reducing and self.sharded:
                    self.reducer.reduce_scalar(self._sumsq)     # the shards' shares of sum(g^2): one 4-byte all-reduce
            self._reducing = reducing
            self.update_due = True
        return loss.detach()

    def apply_update(self, lr_scale: Optional[float] = None, lrs=None, wds=None) -> None:
        """global-norm clip + fused AdamW over the arena (+ bf16 shadow refresh).  ``lrs`` / ``wds``: explicit per-group values
        (an external LR scheduler writing ``param_groups[i]["lr"]``); default: this trainer's cosine schedule."""
        assert self.update_due, "apply_update() before the last micro-batch of the step"
        # under bf16 data parallelism the averaged gradients live in the bf16 communication copy
        self.opt.step(self.lr_scale() if lr_scale is None else lr_scale, sumsq=self._sumsq,
                      grads=self.reducer.result_arena if self._reducing else None, lrs=lrs, wds=wds,
                      grad_scale=self.reducer.grad_scale if self._reducing else 1.0)
        if self.sharded and self._reducing:
            self.reducer.gather_params(overlap=self.gather_overlap)
        self.update_due = False
        self.global_step += 1

    def consolidate(self) -> None:
        """sharded optimizer step: bring every rank's fp32 masters up to date (all ranks call it) — before state_dict() /
        save_pretrained() / evaluation code that reads parameters directly.  A no-op otherwise."""
        self.store.wait_pending()
        if self.sharded and (self.reducer.world > 1 or self.reducer.force):
            self.reducer.gather_masters()
