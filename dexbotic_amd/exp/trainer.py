"""``NativeDexboticTrainer``: the reference's ``DexboticTrainer(transformers.Trainer)`` (dexbotic/exp/trainer.py:18-138) for
the native policy — what an exp script gets instead of ``DexboticTrainer`` when the native backend is opted in.

Same constructor (``exp_config=`` plus the HF ``Trainer`` arguments), same ``_link_exp_config`` mapping from the exp's
``trainer_config`` / ``optimizer_config`` to ``TrainingArguments``, same ``*_loss`` logging; HF's loop (dataloader, LR
scheduler, logging, checkpoint cadence, callbacks) stays in charge.  What changes underneath:

  * ``create_optimizer`` takes the groups ``OptimizerConfig._get_optimizer_grouped_parameters`` builds (base_exp.py:95-203; the
    reference's own function runs unmodified on the native model) and hands them to ``ArenaAdamW``: a ``torch.optim.Optimizer``
    whose ``step()`` is ONE fused launch over the arena (engine.FusedAdamW: global-norm clip on the device + AdamW + bf16 shadow
    refresh).  HF's scheduler drives ``param_groups[i]["lr"]`` as usual.
  * ``training_step`` = ``trainer.NativeTrainer.micro_step``: forward + backward into the gradient arenas, data-parallel
    exchange by engine.GradReducer (not DDP: the weight gradients are written by the dW kernels' epilogues, autograd never
    sees them), sum of squares folded under the backward.  ``max_grad_norm`` (1.0, trainer.py:122) is applied inside the fused
    step, so HF's own ``clip_grad_norm_`` pass is switched off (``TrainingArguments.max_grad_norm = 0``).
  * ``deepspeed`` of the exp config is NOT forwarded: the optimizer state is not sharded (144 GB resident, engine.py header).
    ``gradient_checkpointing`` (reference default True, there to fit 80 GB parts) is forwarded only with
    ``DEXBOTIC_AMD_GRAD_CHECKPOINTING=1``: activations stay resident by default (25 GB of 288 at the CogACT batch); with the
    switch HF's loop calls ``model.gradient_checkpointing_enable()`` and the layer Functions recompute (functional.py).
  * integer inputs stay on the host (``_prepare_inputs``): the splice plan is host arithmetic (splice.py).
  * ``gradient_accumulation_steps`` (2 in the reference recipe, there to fit 80 GB parts) is honoured as a GROUPING: HF hands
    over micro-batch after micro-batch, the core holds them and runs the group as one pass over the concatenated batch
    (trainer.NativeTrainer ``coalesce_micro_batches``: same mean loss, same gradients up to fp32 summation order, the one-pass
    rate).  ``training_step`` then returns 0 for the held micro-batches and the group's sum for the last, so HF's running loss
    is unchanged.  DEXBOTIC_AMD_COALESCE=0 runs pass by pass.

The unmodified ``DexboticTrainer`` / plain HF ``Trainer`` + ``torch.optim.AdamW`` also train the native model on ONE GPU
(ParamStore.external_prelude: gradients re-attached, stale bf16 shadows re-derived) — slower (a foreach AdamW over ~800 arena
views + an 8 B-element shadow cast per step); see INTEGRATION.md.
"""
from __future__ import annotations

import os
from typing import Any, Dict, List, Optional

import torch
from transformers import Trainer, TrainingArguments

from ..data.feeder import HOST_KEYS
from ..engine import OptimConfig
from ..trainer import NativeTrainer


class ArenaAdamW(torch.optim.Optimizer):
    """torch.optim.Optimizer facade over engine.FusedAdamW.  ``param_groups`` are the reference's groups (lr / weight_decay
    per group, mutable by LR schedulers); ``step()`` runs the fused clip + AdamW launch of the owning NativeTrainer."""

    def __init__(self, params, core: NativeTrainer, lr: float = 2e-5, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.core = core
        assert len(self.param_groups) == len(core.opt.group_keys), "core was built from other parameter groups"

    def step(self, closure=None):
        assert closure is None, "ArenaAdamW takes no closure"
        core = self.core
        if not core.update_due and (core._held or core.micro % core.grad_accum != 0):
            # HF closed a SHORT accumulation group (last batches of an epoch whose length does not divide by
            # gradient_accumulation_steps) without telling training_step its size — transformers 4.51, which the reference
            # pins, has no Trainer.current_gradient_accumulation_steps: whatever is held or half-accumulated IS the group.
            # Two consequences, both confined to that last short group of an epoch under 4.51: the held micro-batches'
            # forward / backward run HERE, after training_step already returned 0 for them, so HF's logged tr_loss misses their
            # share (the update itself is exact: test_hf_trainer_gpu's short-group case); and an out-of-memory error in that pass
            # surfaces from optimizer.step() — close_short_group restores grad_accum / micro in its finally block, the step's
            # gradient arena is left as the next begin_step() finds and overwrites it.
            core.close_short_group()
        with torch.no_grad():
            self._apply()

    def _apply(self):
        self.core.apply_update(lrs=[float(g["lr"]) for g in self.param_groups],
                               wds=[float(g["weight_decay"]) for g in self.param_groups])

    def zero_grad(self, set_to_none: bool = True):
        # gradients are arena views overwritten by the next backward's first write (ParamStore.begin_step)
        self.core.store.external_zero_grad()

    def state_dict(self) -> Dict[str, Any]:
        sd = super().state_dict()
        self.core.synchronize()
        # (sharded optimizer step: m / v are this rank's shard, packed — HF saves the optimizer state per rank in that case, like
        #  the reference's ZeRO checkpoints; ``ranges`` makes a shard loaded into another layout fail loudly)
        full = getattr(self.core, "_gathered_moments", None)
        if full is not None:                      # NativeDexboticTrainer._save_checkpoint gathered arena-shaped moments on every rank
            sd["arena"] = {"m": full[0], "v": full[1], "step": self.core.opt.step_count, "ranges": None}
        else:
            sd["arena"] = {"m": self.core.opt.m, "v": self.core.opt.v, "step": self.core.opt.step_count,
                           "ranges": self.core.opt.ranges}
        return sd

    def load_state_dict(self, state_dict) -> None:
        arena = state_dict.get("arena")
        super().load_state_dict({k: v for k, v in state_dict.items() if k != "arena"})
        if arena is not None:
            # (accelerate round-trips the state dict in place: load_moments skips the copy then, and resets the sparse-table state)
            self.core.opt.load_moments(arena["m"], arena["v"], int(arena["step"]))


def link_exp_config(exp_config, **overrides) -> TrainingArguments:
    """trainer_config / optimizer_config -> TrainingArguments, the mapping of DexboticTrainer._link_exp_config
    (trainer.py:88-124), minus what the native backend does itself (module docstring)."""
    tc, oc = exp_config.trainer_config, exp_config.optimizer_config
    args = dict(output_dir=tc.output_dir, num_train_epochs=tc.num_train_epochs, max_steps=tc.num_train_steps,
                per_device_train_batch_size=tc.per_device_train_batch_size,
                gradient_accumulation_steps=tc.gradient_accumulation_steps, save_strategy=tc.save_strategy,
                save_steps=tc.save_steps, save_total_limit=tc.save_total_limit, save_only_model=tc.save_only_model,
                logging_steps=tc.logging_steps, dataloader_num_workers=tc.dataloader_num_workers, bf16=tc.bf16,
                lr_scheduler_type=tc.lr_scheduler_type, lr_scheduler_kwargs=getattr(tc, "lr_scheduler_kwargs", {}) or {},
                run_name=getattr(tc, "run_name", None), remove_unused_columns=False, learning_rate=oc.base_lr,
                adam_beta1=oc.adam_beta1, adam_beta2=oc.adam_beta2, warmup_steps=oc.warmup_steps,
                weight_decay=oc.weight_decay,
                # resident activations unless asked for (module docstring); unsharded optimizer state
                gradient_checkpointing=bool(getattr(tc, "gradient_checkpointing", False)) and
                os.environ.get("DEXBOTIC_AMD_GRAD_CHECKPOINTING", "0") != "0",
                gradient_checkpointing_kwargs={"use_reentrant": False},          # trainer.py:120
                deepspeed=None,
                max_grad_norm=0.0)                                   # the 1.0 clip runs inside the fused optimizer step
    args.update(overrides)
    return TrainingArguments(**args)


class NativeDexboticTrainer(Trainer):
    MAX_GRAD_NORM = 1.0            # DexboticTrainer._link_exp_config: linked_args["max_grad_norm"] = 1.0 (trainer.py:122)

    def __init__(self, *args, **kwargs):
        self.exp_config = kwargs.pop("exp_config")
        self._core: Optional[NativeTrainer] = None
        self._core_kw = dict(kwargs.pop("native", None) or {})
        training_args = kwargs.pop("args", None) or link_exp_config(self.exp_config)
        import types
        # what the reference keeps beside the TrainingArguments (trainer.py: self.added_args): the adapter-only switch
        self.added_args = types.SimpleNamespace(
            tune_mm_mlp_adapter=bool(getattr(self.exp_config.trainer_config, "tune_mm_mlp_adapter", False)))
        super().__init__(*args, args=training_args, **kwargs)
        self.loss_cache: Dict[str, float] = {}
        self._loss_dev: Dict[str, Any] = {}

    # ---- the native machinery behind HF's loop ---------------------------------------------------------------------------
    def _grouped_parameters(self) -> List[dict]:
        if getattr(self, "_grouped", None) is None:
            self._grouped = self.exp_config.optimizer_config._get_optimizer_grouped_parameters(self.model)
        return self._grouped

    @property
    def core(self) -> NativeTrainer:
        if self._core is None:
            oc = self.exp_config.optimizer_config
            cfg = OptimConfig(base_lr=oc.base_lr, weight_decay=oc.weight_decay, adam_beta1=self.args.adam_beta1,
                              adam_beta2=self.args.adam_beta2, adam_epsilon=self.args.adam_epsilon,
                              max_grad_norm=self.MAX_GRAD_NORM)
            name_of = {id(p): n for n, p in self.model.store.params.items()}
            groups = [{"names": [name_of[id(p)] for p in g["params"] if id(p) in name_of],
                       "lr": float(g.get("lr", self.args.learning_rate)), "weight_decay": float(g.get("weight_decay", 0.0))}
                      for g in self._grouped_parameters()]
            # the accumulation micro-batches of an optimizer step run as ONE pass where the model allows it (trainer.NativeTrainer
            # coalesce_micro_batches; DEXBOTIC_AMD_COALESCE=0 or native={"coalesce_micro_batches": False}: pass by pass)
            kw = dict(coalesce_micro_batches=os.environ.get("DEXBOTIC_AMD_COALESCE", "1") != "0")
            kw.update(self._core_kw)
            self._core = NativeTrainer(self.model, cfg, grad_accum=self.args.gradient_accumulation_steps,
                                       optimizer_groups=groups, **kw)
        return self._core

    def create_optimizer(self) -> torch.optim.Optimizer:
        if self.optimizer is None:
            self.optimizer = ArenaAdamW(self._grouped_parameters(), self.core, lr=self.args.learning_rate,
                                        betas=(self.args.adam_beta1, self.args.adam_beta2), eps=self.args.adam_epsilon,
                                        weight_decay=self.args.weight_decay)
        return self.optimizer

    def create_accelerator_and_postprocess(self):
        super().create_accelerator_and_postprocess()
        orig = self.accelerator.prepare_model
        from ..model.dexbotic_arch import NativePreTrainedMixin

        def prepare_model(model, *a, **k):
            if isinstance(model, NativePreTrainedMixin):
                return model              # no DDP wrapper: engine.GradReducer averages the gradient arenas
            return orig(model, *a, **k)
        self.accelerator.prepare_model = prepare_model

    def _wrap_model(self, model, training=True, dataloader=None):
        return model

    def _prepare_inputs(self, inputs):
        host = {k: inputs[k] for k in HOST_KEYS if k in inputs and torch.is_tensor(inputs[k]) and not inputs[k].is_cuda}
        rest = super()._prepare_inputs({k: v for k, v in inputs.items() if k not in host})
        rest.update(host)
        return rest

    def training_step(self, model, inputs, num_items_in_batch=None):
        model.train()
        inputs = self._prepare_inputs(inputs)
        core = self.core
        if core.micro % core.grad_accum == 0:
            # first micro-batch of an accumulation group: HF closes a SHORT group at the end of an epoch whose length does not
            # divide by gradient_accumulation_steps (Trainer._inner_training_loop: ``remainder``, do_sync_step on the last
            # batch) and calls optimizer.step() after it — the group's real size is len(batch_samples), published as
            # current_gradient_accumulation_steps
            group = int(getattr(self, "current_gradient_accumulation_steps", None) or self.args.gradient_accumulation_steps)
            if group != core.grad_accum:
                core.set_grad_accum(group)
        # HF does NOT divide the loss by the accumulation steps for a model whose forward takes **kwargs once it passes
        # num_items_in_batch (Trainer.training_step; true of the reference's forwards under its pinned transformers 4.51): the
        # micro-batch gradients are then summed, not averaged, before the 1.0 clip.  Mirrored here.
        summed = bool(getattr(self, "model_accepts_loss_kwargs", False)) and num_items_in_batch is not None
        loss = core.micro_step(inputs, loss_scale=1.0 if summed else None)
        if core.last_output is not None:              # (a coalesced group reports with its last micro-batch)
            self._cache_losses(core.last_output)
        accum = core.grad_accum
        return loss if (summed or accum == 1) else loss / accum

    def compute_loss(self, model, inputs, return_outputs=False, *args, **kwargs):
        """evaluation-side path (no backward): same *_loss bookkeeping as the reference (trainer.py:126-134)"""
        loss, outputs = super().compute_loss(model, inputs, return_outputs=True)
        self._cache_losses(outputs)
        return (loss, outputs) if return_outputs else loss

    def _cache_losses(self, outputs) -> None:
        # detached DEVICE scalars: converting here would drain the stream once per micro-batch (the step itself never waits for
        # the host); log() converts, every logging_steps
        for key in [k for k in outputs.keys() if k.endswith("_loss")]:
            val = outputs[key]
            self._loss_dev[key] = None if val is None else (val.detach() if torch.is_tensor(val) else val)

    def log(self, logs: Dict[str, float], start_time: Optional[float] = None) -> None:
        for key, val in self._loss_dev.items():
            f = 0.0 if val is None else float(val)
            if f == 0.0:                              # the reference keeps the last non-zero value (trainer.py:126-134)
                self.loss_cache.setdefault(key, 0.0)
            else:
                self.loss_cache[key] = f
        logs.update(self.loss_cache)
        super().log(logs, start_time)

    # ---- checkpoints: HF's cadence, the reference's contents (dexbotic/exp/trainer.py:38-87) -------------------------------
    def _save(self, output_dir: Optional[str] = None, state_dict=None) -> None:
        """what ``Trainer._save`` writes for a PreTrainedModel — config.json + model.safetensors + the tokenizer — through the
        native model's own ``save_pretrained`` (it is not a ``PreTrainedModel``, so HF would write a bare state dict that
        ``from_pretrained`` cannot read back).  Adapter-only runs save nothing here, like the reference."""
        if getattr(getattr(self, "added_args", None), "tune_mm_mlp_adapter", False):
            return
        import os
        output_dir = output_dir if output_dir is not None else self.args.output_dir
        os.makedirs(output_dir, exist_ok=True)
        self.core.synchronize()                       # an overlapped optimizer update still in flight
        # (sharded optimizer step: _save_checkpoint gathered the fp32 masters on every rank before this rank-0 write)
        self.model.save_pretrained(output_dir)
        tok = getattr(self, "processing_class", None) or getattr(self, "tokenizer", None)
        if tok is not None and hasattr(tok, "save_pretrained"):
            tok.save_pretrained(output_dir)
        torch.save(self.args, os.path.join(output_dir, "training_args.bin"))

    def _save_checkpoint(self, model, trial, metrics=None) -> None:
        import os
        from transformers.trainer_utils import PREFIX_CHECKPOINT_DIR
        output_dir = os.path.join(self._get_output_dir(trial=trial), f"{PREFIX_CHECKPOINT_DIR}-{self.state.global_step}")
        main = self.args.local_rank in (0, -1)
        # sharded optimizer step (trainer.NativeTrainer shard_optimizer, the default under data parallelism): a rank's fp32 masters
        # are current only for the shard it owns — gathered here, on EVERY rank (HF calls _save_checkpoint on all of them), as the
        # reference's ZeRO-3 save does (dexbotic/exp/trainer.py:145-189)
        self.core.consolidate()
        if self.core.sharded and not self.args.save_only_model:
            # HF writes optimizer.pt from rank 0 only: the moments are gathered into arena-shaped tensors first (all ranks take part),
            # so that the checkpoint resumes on any world size (FusedAdamW.load_moments keeps a rank's own shard)
            self.core._gathered_moments = self.core.opt.full_moments(self.core.reducer)
        try:
            self._save_checkpoint_body(model, trial, metrics, output_dir, main)
        finally:
            self.core._gathered_moments = None

    def _save_checkpoint_body(self, model, trial, metrics, output_dir, main) -> None:
        import os
        if getattr(getattr(self, "added_args", None), "tune_mm_mlp_adapter", False):
            # only the projector (trainer.py:41-57): config.json + mm_projector.bin
            if main:
                os.makedirs(output_dir, exist_ok=True)
                self.core.synchronize()
                self.model.config.save_pretrained(output_dir)
                weights = {k: v.detach().cpu().clone() for k, v in self.model.state_dict().items() if "mm_projector" in k}
                torch.save(weights, os.path.join(output_dir, "mm_projector.bin"))
            return
        # transformers 4.x: _save_checkpoint(model, trial, metrics=None); 5.x dropped ``metrics`` — picked from the signature, not
        # by catching TypeError around the whole save (that would re-run a half-written save and hide the real error)
        import inspect
        if "metrics" in inspect.signature(Trainer._save_checkpoint).parameters:
            super()._save_checkpoint(model, trial, metrics)
        else:
            super()._save_checkpoint(model, trial)
        if main:
            self._copy_norm_stats_to_checkpoint(output_dir)

    def _copy_norm_stats_to_checkpoint(self, checkpoint_dir: str) -> None:
        """norm_stats.json travels with every checkpoint (trainer.py:68-82): inference de-normalises with it"""
        import os
        import shutil
        src = os.path.join(self.args.output_dir, "norm_stats.json")
        if os.path.exists(src) and os.path.isdir(checkpoint_dir):
            shutil.copy2(src, os.path.join(checkpoint_dir, "norm_stats.json"))
