"""Activation registry (mirror of adell_mri/modules/activations.py:6-31).

The values are the stock ``torch.nn`` classes, exactly as in the reference, so
``activation_fn`` arguments and YAML strings interchange; ``act_spec`` maps an
instantiated activation to the id / parameters the fused HIP kernel takes.
"""
import torch

activation_factory = {
    "identity": torch.nn.Identity,
    "elu": torch.nn.ELU,
    "hard_shrink": torch.nn.Hardshrink,
    "hard_tanh": torch.nn.Hardtanh,
    "leaky_relu": torch.nn.LeakyReLU,
    "logsigmoid": torch.nn.LogSigmoid,
    "gelu": torch.nn.GELU,
    "prelu": torch.nn.PReLU,
    "relu": torch.nn.ReLU,
    "relu6": torch.nn.ReLU6,
    "rrelu": torch.nn.RReLU,
    "selu": torch.nn.SELU,
    "celu": torch.nn.CELU,
    "sigmoid": torch.nn.Sigmoid,
    "softplus": torch.nn.Softplus,
    "soft_shrink": torch.nn.Softshrink,
    "softsign": torch.nn.Softsign,
    "tanh": torch.nn.Tanh,
    "tanh_shrink": torch.nn.Tanhshrink,
    "threshold": torch.nn.Threshold,
    "softmin": torch.nn.Softmin,
    "softmax": torch.nn.Softmax,
    "logsoftmax": torch.nn.LogSoftmax,
    "swish": torch.nn.SiLU,
}


def act_spec(module):
    """(act name, scalar parameter, weight tensor or None) for the HIP kernel."""
    m = module
    if m is None or isinstance(m, torch.nn.Identity):
        return "identity", 0.0, None
    if isinstance(m, torch.nn.SiLU):
        return "swish", 0.0, None
    if isinstance(m, torch.nn.ReLU):
        return "relu", 0.0, None
    if isinstance(m, torch.nn.LeakyReLU):
        return "leaky_relu", float(m.negative_slope), None
    if isinstance(m, torch.nn.PReLU):
        return "prelu", 0.0, m.weight
    if isinstance(m, torch.nn.GELU):
        if getattr(m, "approximate", "none") != "none":
            raise NotImplementedError("GELU(approximate='tanh') has no HIP kernel")
        return "gelu", 0.0, None
    if isinstance(m, torch.nn.Sigmoid):
        return "sigmoid", 0.0, None
    if isinstance(m, torch.nn.Tanh):
        return "tanh", 0.0, None
    if isinstance(m, torch.nn.ELU):
        return "elu", float(m.alpha), None
    raise NotImplementedError(
        f"activation {type(m).__name__} has no HIP kernel on the adell_mri_amd path")
