"""The two exp-layer dataclasses the trainer link consumes — field-for-field mirrors of ``OptimizerConfig`` and
``TrainerConfig`` (dexbotic/exp/base_exp.py:64-93, 206-256) — so ``NativeDexboticTrainer`` can be driven without the
reference tree.  A reference ``BaseExp`` object works as ``exp_config`` just as well (duck typing: ``.trainer_config``,
``.optimizer_config`` with ``_get_optimizer_grouped_parameters``).  The rest of the exp layer (model / data / action /
inference configs, launch logic) is out of scope here."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

import torch.nn as nn

from ..engine import no_decay_name


@dataclass
class OptimizerConfig:
    optim: str = "adamw_torch"
    base_lr: float = 2e-5
    weight_decay: float = 0.0
    warmup_ratio: float = 0.03
    warmup_steps: int = 0
    adam_beta1: float = 0.9
    adam_beta2: float = 0.999
    adam_epsilon: float = 1e-8
    mm_projector_lr: Optional[float] = None
    mm_vision_lr: Optional[float] = None
    action_head_lr: Optional[float] = None

    def _get_optimizer_grouped_parameters(self, model: nn.Module) -> List[dict]:
        """the <= 8 groups of base_exp.py:95-203: {mm_projector, mm_vision, action_head} x {decay, no decay} for the modules
        with a learning rate of their own, then base decay / no decay.  No weight decay for parameters of nn.LayerNorm
        modules and for names containing "bias" (tests/test_dropin_reference_exp.py holds the reference's own function,
        run on the native model, to the same groups)."""
        store = getattr(model, "store", None)
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        decay = {n for n, _ in named if not no_decay_name(n, store)}
        out, taken = [], set()
        for lr, prefix in ((self.mm_projector_lr, getattr(model, "mm_projector_prefix", "mm_projector")),
                           (self.mm_vision_lr, getattr(model, "mm_vision_prefix", "mm_vision")),
                           (self.action_head_lr, getattr(model, "action_head_prefix", "action_head"))):
            if lr is None:
                continue
            mine = {n for n, _ in named if prefix in n}
            out.append({"params": [p for n, p in named if n in mine and n in decay], "weight_decay": self.weight_decay, "lr": lr})
            out.append({"params": [p for n, p in named if n in mine and n not in decay], "weight_decay": 0.0, "lr": lr})
            taken |= mine
        out.append({"params": [p for n, p in named if n not in taken and n in decay], "weight_decay": self.weight_decay,
                    "lr": self.base_lr})
        out.append({"params": [p for n, p in named if n not in taken and n not in decay], "weight_decay": 0.0,
                    "lr": self.base_lr})
        return out


@dataclass
class TrainerConfig:
    deepspeed: Optional[str] = None            # reference default './script/deepspeed/zero3.json': not forwarded (exp/trainer.py)
    output_dir: Optional[str] = None
    num_train_epochs: int = 1
    num_train_steps: Optional[int] = -1
    per_device_train_batch_size: int = 8
    gradient_accumulation_steps: int = 2
    save_strategy: str = "steps"
    save_steps: int = 20000
    save_total_limit: int = 1
    save_only_model: bool = True
    logging_steps: int = 10
    wandb_project: str = "dexbotic"
    gradient_checkpointing: bool = True        # honoured with DEXBOTIC_AMD_GRAD_CHECKPOINTING=1 (exp/trainer.py); else resident
    dataloader_num_workers: int = 8
    model_max_length: int = 2048
    debug_mode: bool = False
    bf16: bool = True
    tf32: bool = True
    lr_scheduler_type: str = "cosine"
    lr_scheduler_kwargs: dict = field(default_factory=dict)
    tune_mm_mlp_adapter: bool = False
    run_name: Optional[str] = None


@dataclass
class ExpConfig:
    """just the two members the trainer reads"""
    trainer_config: TrainerConfig = field(default_factory=TrainerConfig)
    optimizer_config: OptimizerConfig = field(default_factory=OptimizerConfig)
