"""Minimal fine-tune driver for the native policy: what HF ``Trainer.training_step`` + ``optimizer.step`` do
for the reference (exp/trainer.py:25-36,88-138; exp/base_exp.py:865-871), on the flat arenas.

    step(batch):  begin_step -> model(**batch).loss.backward() -> [DP all-reduce finishes] ->
                  global-norm clip + fused AdamW (+ bf16 shadow refresh) -> cosine lr

One process per GPU; with ``torch.distributed`` initialised (backend "nccl" = RCCL over xGMI) the gradient
arena is averaged across ranks by engine.GradReducer, overlapped with the backward.
"""
from __future__ import annotations

import os

from typing import Dict, Optional

import torch

from . import kernels as K

from .engine import FusedAdamW, GradNormTracker, GradReducer, OptimConfig, cosine_lr_scale


class NativeTrainer:
    def __init__(self, model, optim: Optional[OptimConfig] = None, total_steps: int = 0, warmup_steps: int = 0,
                 grad_accum: int = 1, distributed: Optional[bool] = None, min_bucket_bytes: int = 256 << 20,
                 force_reducer: bool = False, grad_comm_dtype: torch.dtype = torch.float32, grad_sync: str = "rs_ag",
                 grad_dtype: torch.dtype = torch.float32, overlap_optimizer: bool = False, optimizer_groups=None,
                 native_avg_world1: bool = False):
        import torch.distributed as dist
        self.model = model
        self.store = model.store
        self.cfg = optim or OptimConfig()
        unused = list(model.unused_parameter_names()) if hasattr(model, "unused_parameter_names") else []
        # overlap_optimizer: the fused AdamW runs bucket by bucket in forward order on a side stream and the NEXT step's
        # forward waits per bucket (engine.FusedAdamW(overlap=True)).  step() then returns with the update still in flight:
        # everything that goes through the store's views (the model's own forward / inference, sync_shadow) orders itself;
        # code that reads parameters directly (p.data, state_dict()) calls trainer.synchronize() first.
        # optimizer_groups: explicit parameter groups [{"names": [...]}, ...] (exp/trainer.NativeDexboticTrainer: the groups the
        # exp's OptimizerConfig built); default: engine.FusedAdamW's own name rule.
        self.opt = FusedAdamW(self.store, self.cfg, exclude=unused, overlap=overlap_optimizer, groups=optimizer_groups)
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
                                       comm_dtype=grad_comm_dtype, algo=grad_sync, native_avg_world1=native_avg_world1)
        elif bf16_grads:
            self.reducer = GradReducer(self.store, min_bucket_bytes=min_bucket_bytes, skip=unused, comm_dtype=torch.bfloat16,
                                       local_only=True)
        self.store.bf16_grads = bf16_grads
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
        self.update_due = False
        self._sumsq, self._reducing = None, False
        self.last_output = None

    def set_grad_accum(self, n: int) -> None:
        """change the number of micro-batches per optimizer step (between optimizer steps only)"""
        assert not self.update_due and self.micro % self.grad_accum == 0, "set_grad_accum() in the middle of an optimizer step"
        assert n == 1 or not self.store.bf16_grads, "bf16 gradient arena needs one micro-batch per step"
        self.grad_accum, self.micro = int(n), 0
        local = self.reducer is None or self.reducer.local_only
        self.store.epi_sumsq = local and self.norm_tracker is not None

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
        ``loss_scale``: factor on the loss before backward; default 1 / grad_accum (the mean over micro-batches)."""
        first = self.micro % self.grad_accum == 0
        last = (self.micro + 1) % self.grad_accum == 0
        self.store.last_micro = last
        # exactly two micro-batches: dW of the linears = one product over both (the pair of micro-batch 1 is held until 2)
        self.store.accum_merge = self.grad_accum == 2 and self.store.device.type == "cuda" and not self.store.bf16_grads \
            and os.environ.get("DXA_NO_ACCUM_MERGE") is None
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
        with K.f32_gemm_mode(getattr(self.model.config, "fp32_matmul", "exact")):
            out = self.model(**batch)
            loss = out.loss
            self.store.wait_pending()           # (overlapped optimizer: buckets the forward never touched)
            scale = (1.0 / self.grad_accum) if loss_scale is None else float(loss_scale)
            (loss * scale if scale != 1.0 else loss).backward()
        self.last_output = out
        self.store.flush_wgrads()               # (only when autograd pruned a consumer of a multiply-used parameter)
        if last and self.store._accum_stash:
            from .functional import flush_accum
            flush_accum(self.store)             # (a parameter the last micro-batch did not use)
        self.micro += 1
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
            self._reducing = reducing
            self.update_due = True
        return loss.detach()

    def apply_update(self, lr_scale: Optional[float] = None, lrs=None, wds=None) -> None:
        """global-norm clip + fused AdamW over the arena (+ bf16 shadow refresh).  ``lrs`` / ``wds``: explicit per-group values
        (an external LR scheduler writing ``param_groups[i]["lr"]``); default: this trainer's cosine schedule."""
        assert self.update_due, "apply_update() before the last micro-batch of the step"
        # under bf16 data parallelism the averaged gradients live in the bf16 communication copy
        self.opt.step(self.lr_scale() if lr_scale is None else lr_scale, sumsq=self._sumsq,
                      grads=self.reducer.result_arena if self._reducing else None, lrs=lrs, wds=wds)
        self.update_due = False
        self.global_step += 1
