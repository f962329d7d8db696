"""dexbotic_amd — MI355X-native (gfx950 / CDNA4) drop-in for the DB-CogACT hot path of dexmal/dexbotic.

Importing the package loads libdexbotic_amd.so through ctypes and fails loudly if it is missing:
there is no CPU or eager-PyTorch fallback for the compute path.

    from dexbotic_amd.model.cogact.cogact_arch import CogActConfig, CogACTForCausalLM

mirrors ``dexbotic.model.cogact.cogact_arch`` (same class names, ``forward`` / ``inference_action``
contracts, ``state_dict`` keys and ``model_type`` registry strings).
"""
import sys as _sys


def _is_build_invocation() -> bool:
    """True only for ``python -m dexbotic_amd.build`` — the one command that must work BEFORE the library exists (the
    ImportError below names it).  runpy imports this package before it runs the build module, so the library check is
    skipped for exactly that command line and for nothing else."""
    oa = list(getattr(_sys, "orig_argv", []))
    return any(a == "-m" and i + 1 < len(oa) and oa[i + 1] == "dexbotic_amd.build" for i, a in enumerate(oa))


if not _is_build_invocation():
    from . import _lib  # noqa: F401  (raises ImportError when the native library is absent)

__version__ = "0.1.0"


def model_registry():
    """``config.model_type`` -> (Config, ForCausalLM): the HF AutoConfig/AutoModel registry strings of the
    reference (dexbotic_arch.py:18, cogact_arch.py:14)."""
    from .model.cogact.cogact_arch import CogActConfig, CogACTForCausalLM
    from .model.dexbotic_arch import DexboticConfig, DexboticForCausalLM
    from .model.memvla.memvla_arch import MemVLAConfig, MemVLAForCausalLM
    from .model.pi0.pi0_arch import Pi0Config, Pi0ForCausalLM
    return {"dexbotic": (DexboticConfig, DexboticForCausalLM),
            "dexbotic_cogact": (CogActConfig, CogACTForCausalLM),
            "dexbotic_pi0": (Pi0Config, Pi0ForCausalLM),
            "dexbotic_memvla": (MemVLAConfig, MemVLAForCausalLM)}


def hybrid_cogact():
    """HybridCogACTForCausalLM shares model_type "dexbotic_cogact" with CogACT (hybrid_cogact_arch.py:18-22)"""
    from .model.cogact.hybrid_cogact_arch import HybridCogACTForCausalLM
    return HybridCogACTForCausalLM


def discrete_vla():
    """DiscreteVLAForCausalLM shares model_type "dexbotic" with the base class in the reference
    (discrete_vla_arch.py:12-13); exps pick it by class, so it is exported by name here."""
    from .model.discrete_vla.discrete_vla_arch import DiscreteVLAForCausalLM
    return DiscreteVLAForCausalLM


def from_pretrained(path: str, **kw):
    """load any registered policy from a reference-format checkpoint directory"""
    import json
    import os
    with open(os.path.join(path, "config.json")) as f:
        mt = json.load(f).get("model_type", "dexbotic")
    _, cls = model_registry()[mt]
    return cls.from_pretrained(path, **kw)
