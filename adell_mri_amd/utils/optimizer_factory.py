"""Optimiser factory (mirror of adell_mri/utils/optimizer_factory.py:1-54, imported by
entrypoints/segmentation/train.py:30): the same three names -- ``OPTIMIZER_MATCH``,
``get_optimizer``, ``optimizer_eps_from_precision`` -- with the eight optimiser names mapped to the
fused flat-buffer optimisers of ``adell_mri_amd.optim`` (one HIP launch per step over one flat
parameter buffer) instead of ``torch.optim``."""
from ..optim import (FusedAdagrad, FusedAdam, FusedAdamax, FusedAdamW, FusedNAdam, FusedRAdam,
                     FusedRMSprop, FusedSGD)

OPTIMIZER_EPS_DEFAULT = 1e-8

OPTIMIZER_MATCH = {
    "adam": FusedAdam,
    "adamw": FusedAdamW,
    "adamax": FusedAdamax,
    "sgd": FusedSGD,
    "adagrad": FusedAdagrad,
    "nadam": FusedNAdam,
    "radam": FusedRAdam,
    "rmsprop": FusedRMSprop,
}


def get_optimizer(optimizer_str: str, *args, **kwargs):
    """The optimiser a name stands for, built with ``args`` / ``kwargs``; as in the reference
    (optimizer_factory.py:17-30) an unknown name returns ``None``."""
    if optimizer_str in OPTIMIZER_MATCH:
        return OPTIMIZER_MATCH[optimizer_str](*args, **kwargs)


def optimizer_eps_from_precision(precision: str) -> float:
    """Optimiser epsilon for a training precision string (optimizer_factory.py:33-54): 1e-4 for
    pure float16 (``"16-true"``), the default 1e-8 for everything else, ``None`` included."""
    if precision is None:
        return OPTIMIZER_EPS_DEFAULT
    if str(precision).lower() == "16-true":
        return 1e-4
    return OPTIMIZER_EPS_DEFAULT
