"""Drop-in check of the plugin surface: the REFERENCE's own experiment-layer code —
``CogACTModelConfig._freeze_model`` (dexbotic/exp/cogact_exp.py:106-124) and
``OptimizerConfig._get_optimizer_grouped_parameters`` (dexbotic/exp/base_exp.py:95-203) — is imported from
/root/reference and run, unmodified, on the NATIVE model; the parameter groups it builds (which parameters, which learning
rate, weight decay or not — selected by module TYPE through transformers' get_parameter_names(ALL_LAYERNORM_LAYERS)) must
be the groups engine.FusedAdamW uses.  Third-party packages the reference's exp layer imports but this container lacks
(loguru, megfile, easydict, numpydantic, decord, av, cv2, timm, ...) are replaced by inert stand-ins: none of them is
touched by the two functions under test.  Skipped where /root/reference does not exist (the GPU box)."""
import importlib.abc
import importlib.machinery
import os
import sys
import types

import pytest
import torch

from tests.helpers import CFGS, build_product
from oracle.weights import cogact_shapes, make_weights

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "dexbotic")), reason="needs /root/reference")

MISSING = ("loguru", "megfile", "easydict", "numpydantic", "decord", "av", "cv2", "timm", "albumentations", "deepspeed",
           "wandb", "peft", "flask", "tyro", "imageio", "h5py", "openpi_client")


class _Meta(type):
    """stand-in classes answer any CLASS attribute too (enum members, constants) and subscript like generics"""

    def __getattr__(cls, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _make(name)

    def __getitem__(cls, item):
        return cls

    def __or__(cls, other):
        return cls

    def __ror__(cls, other):
        return cls


def _make(name):
    return _Meta(name, (), {"__init__": lambda self, *a, **k: None, "__call__": lambda self, *a, **k: None,
                            "__getattr__": lambda self, k: (lambda *a, **kw: None)})


class _Anything(types.ModuleType):
    """module stand-in: every attribute is a permissive class (usable as base class, decorator, callable, annotation)"""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        val = _make(name)
        setattr(self, name, val)
        return val


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".")[0] in MISSING:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _Anything(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


@pytest.fixture(scope="module")
def ref_exp():
    import transformers  # noqa: F401  (first: its lazy importer probes optional packages by spec)
    finder = _StubFinder()
    sys.meta_path.append(finder)
    sys.path.insert(0, REF)
    saved = {k: v for k, v in sys.modules.items() if k == "dexbotic" or k.startswith("dexbotic.")}
    for k in saved:
        del sys.modules[k]
    # pydantic cannot build a schema for a stand-in NDArray: the norm-stats module is not on the path under test
    norm = types.ModuleType("dexbotic.data.utils.normalize")
    norm.RunningStats = norm.NormStats = object
    norm.save = norm.load = lambda *a, **k: None
    try:
        import dexbotic.data.utils as du
        sys.modules["dexbotic.data.utils.normalize"] = norm
        du.normalize = norm
        from dexbotic.exp import cogact_exp
        yield cogact_exp
    finally:
        sys.meta_path.remove(finder)
        sys.path.remove(REF)
        for k in [k for k in sys.modules if k == "dexbotic" or k.startswith("dexbotic.") or k.split(".")[0] in MISSING]:
            del sys.modules[k]
        sys.modules.update(saved)


def _native_model():
    cfg = CFGS["t1"]
    return build_product(cfg, make_weights(cogact_shapes(cfg), 3), "float32", "cpu", train=True)


@pytest.mark.parametrize("lrs", [dict(), dict(mm_projector_lr=1e-4, mm_vision_lr=2e-6, action_head_lr=5e-5)])
@pytest.mark.parametrize("freeze", [dict(), dict(freeze_mm_vision=True), dict(freeze_llm=True, freeze_action_head=True)])
def test_reference_freeze_and_optimizer_groups_on_native_model(ref_exp, lrs, freeze):
    from dexbotic_amd.engine import FusedAdamW, OptimConfig
    m = _native_model()
    mc = ref_exp.CogACTModelConfig.__new__(ref_exp.CogACTModelConfig)        # dataclass fields only; no checkpoint I/O
    for k in ("freeze_llm", "freeze_mm_projector", "freeze_mm_vision", "freeze_action_head"):
        setattr(mc, k, freeze.get(k, False))
    mc._freeze_model(m)                                                       # reference code, native model
    frozen = {n for n, p in m.named_parameters() if not p.requires_grad}
    if freeze.get("freeze_mm_vision"):
        assert frozen and all(".mm_vision_tower." in n for n in frozen)
    oc = ref_exp.CogACTOptimizerConfig(base_lr=2e-5, weight_decay=0.1, **lrs)
    groups = oc._get_optimizer_grouped_parameters(m.model)                    # reference code, native model
    by_param = {}
    for g in groups:
        for p in g["params"]:
            assert id(p) not in by_param, "parameter in two groups"
            by_param[id(p)] = (g["lr"], g["weight_decay"])
    opt = FusedAdamW(m.store, OptimConfig(base_lr=2e-5, weight_decay=0.1, **lrs),
                     exclude=[n for n in m.store.slots if not m.store.params[n].requires_grad] + ["lm_head.weight"])
    lr_of = {"base": 2e-5, "mm_projector": lrs.get("mm_projector_lr"), "mm_vision": lrs.get("mm_vision_lr"),
             "action_head": lrs.get("action_head_lr")}
    seen = 0
    for name, p in m.model.named_parameters():
        full = "model." + name
        if not p.requires_grad:
            assert id(p) not in by_param and full not in opt.group_of
            continue
        lr_key, decayed = opt.group_of[full]
        assert by_param[id(p)] == (lr_of[lr_key], 0.1 if decayed else 0.0), full
        seen += 1
    assert seen == len(by_param) and seen > 0


def test_native_trainer_takes_the_reference_exp_config_objects(ref_exp, tmp_path):
    """NativeDexboticTrainer (the opt-in replacement of DexboticTrainer) constructed from the REFERENCE's own config objects —
    dexbotic.exp.base_exp.TrainerConfig with its defaults (deepspeed zero3.json, gradient_checkpointing=True, bf16, 8 x accum 2,
    cosine) and CogACTOptimizerConfig with per-module learning rates — on the native model: the linked TrainingArguments, and
    the fused optimizer's groups = the groups the reference's _get_optimizer_grouped_parameters returns for the *ForCausalLM
    (what DexboticTrainer.create_optimizer passes, trainer.py:25-36).  CPU: construction only, no step."""
    import types as _t
    from dexbotic.exp import base_exp
    from dexbotic_amd.exp.trainer import ArenaAdamW, NativeDexboticTrainer, link_exp_config
    tc = base_exp.TrainerConfig(output_dir=str(tmp_path))
    assert tc.gradient_checkpointing and tc.deepspeed and tc.gradient_accumulation_steps == 2
    oc = ref_exp.CogACTOptimizerConfig(base_lr=2e-5, weight_decay=0.1, mm_projector_lr=1e-4, action_head_lr=5e-5)
    exp = _t.SimpleNamespace(trainer_config=tc, optimizer_config=oc)
    args = link_exp_config(exp, use_cpu=True, bf16=False, report_to=[])
    assert args.gradient_accumulation_steps == 2 and args.per_device_train_batch_size == 8 and args.learning_rate == 2e-5
    assert args.lr_scheduler_type == "cosine" and args.max_grad_norm == 0.0            # the 1.0 clip runs inside the fused step
    assert not args.gradient_checkpointing and args.deepspeed is None                  # deliberately not forwarded
    m = _native_model()
    tr = NativeDexboticTrainer(model=m, args=args, train_dataset=[0] * 4, exp_config=exp)
    opt = tr.create_optimizer()
    assert isinstance(opt, ArenaAdamW) and tr.core.cfg.max_grad_norm == 1.0 and tr.core.grad_accum == 2
    ref_groups = oc._get_optimizer_grouped_parameters(m)                               # reference code on the *ForCausalLM
    assert len(opt.param_groups) == len(ref_groups) == len(tr.core.opt.group_keys) == 6
    name_of = {id(p): n for n, p in m.named_parameters()}
    unused = set(m.unused_parameter_names())
    for gi, g in enumerate(ref_groups):
        assert opt.param_groups[gi]["lr"] == g["lr"] and opt.param_groups[gi]["weight_decay"] == g["weight_decay"]
        for p in g["params"]:
            n = name_of[id(p)]
            if n in unused:
                assert n not in tr.core.opt.group_of                                   # never gets a gradient: not updated
            else:
                assert tr.core.opt.group_of[n] == tr.core.opt.group_keys[gi], n
