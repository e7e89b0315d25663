"""Plugin registry with the reference's decorator names and semantics.

The reference registers its ASR components into fairseq's registries by decorator
(fairseq/tasks/__init__.py register_task, fairseq/models/__init__.py:110-170 register_model /
register_model_architecture, fairseq/criterions/__init__.py register_criterion,
fairseq/optim/lr_scheduler/__init__.py register_lr_scheduler,
fairseq/data/audio/feature_transforms/__init__.py register_audio_feature_transform) and resolves
them by string name from configs.  fairseq is not importable on the GPU box, so the same
decorators are provided here; when fairseq *is* importable `mirror_into_fairseq()` also inserts
the classes into fairseq's registries so `--user-dir`-style drop-in works.
"""
from typing import Callable, Dict

TASK_REGISTRY: Dict[str, type] = {}
TASK_DATACLASS_REGISTRY: Dict[str, type] = {}
MODEL_REGISTRY: Dict[str, type] = {}
MODEL_DATACLASS_REGISTRY: Dict[str, type] = {}
ARCH_MODEL_REGISTRY: Dict[str, type] = {}
ARCH_CONFIG_REGISTRY: Dict[str, Callable] = {}
CRITERION_REGISTRY: Dict[str, type] = {}
CRITERION_DATACLASS_REGISTRY: Dict[str, type] = {}
LR_SCHEDULER_REGISTRY: Dict[str, type] = {}
OPTIMIZER_REGISTRY: Dict[str, type] = {}
AUDIO_FEATURE_TRANSFORM_REGISTRY: Dict[str, type] = {}


def _make(reg, dc_reg, kind):
    def register(name, dataclass=None):
        def deco(cls):
            if name in reg:
                raise ValueError(f"Cannot register duplicate {kind} ({name})")
            reg[name] = cls
            cls.__dataclass = dataclass
            if dataclass is not None and dc_reg is not None:
                dc_reg[name] = dataclass
            return cls

        return deco

    return register


register_task = _make(TASK_REGISTRY, TASK_DATACLASS_REGISTRY, "task")
register_model = _make(MODEL_REGISTRY, MODEL_DATACLASS_REGISTRY, "model")
register_criterion = _make(CRITERION_REGISTRY, CRITERION_DATACLASS_REGISTRY, "criterion")
register_lr_scheduler = _make(LR_SCHEDULER_REGISTRY, None, "lr scheduler")
register_optimizer = _make(OPTIMIZER_REGISTRY, None, "optimizer")


def register_audio_feature_transform(name):
    def deco(cls):
        if name in AUDIO_FEATURE_TRANSFORM_REGISTRY:
            raise ValueError(f"Cannot register duplicate transform ({name})")
        AUDIO_FEATURE_TRANSFORM_REGISTRY[name] = cls
        return cls

    return deco


def register_model_architecture(model_name, arch_name):
    def deco(fn):
        if model_name not in MODEL_REGISTRY:
            raise ValueError(f"Cannot register model architecture for unknown model type ({model_name})")
        ARCH_MODEL_REGISTRY[arch_name] = MODEL_REGISTRY[model_name]
        ARCH_CONFIG_REGISTRY[arch_name] = fn
        return fn

    return deco


def mirror_into_fairseq():
    """Insert every espresso_amd component into fairseq's own registries behind fairseq's call contracts (needs fairseq and
    espresso importable): see espresso_amd/fairseq_plugin.py; pinned by tests/test_fairseq_binding.py."""
    from . import criterions, models, tasks  # noqa: F401  (make sure every component is registered here first)
    from .fairseq_plugin import install

    return install()
