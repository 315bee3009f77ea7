"""Name-keyed deterministic parameter values (TEST INFRASTRUCTURE ONLY).

A parameter's values depend only on its ``state_dict`` key and shape, never on
module construction order, so the real reference (in oracle/make_golden.py), the
torch oracle and the HIP product can all be loaded with identical weights
without shipping them.
"""
import zlib

import numpy as np


def tensor_for(name, shape, scale=None):
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    shape = tuple(int(s) for s in shape)
    if scale is None:
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else 16
        scale = 1.0 / np.sqrt(max(fan_in, 1))
    return rng.uniform(-scale, scale, size=shape).astype(np.float32)


def fill_state_dict(sd, gain=1.0, norm_weight_offset=0.0):
    """Every floating-point entry of a torch state_dict gets tensor_for(key); matrices and
    kernels (dim > 1) are multiplied by ``gain`` (deep stacks of Linear layers need > 1 to
    keep the signal above the biases). ``norm_weight_offset``: added to every 1-D ``*.weight``
    (the LayerNorm / BatchNorm scales): 1.0 centres them where their real initialisation sits --
    with scales of +-0.25 the biases drown the differences between batch items and a VICReg
    fixture collapses (batch std 4e-3, covariance term 5e-8)."""
    import torch

    out = {}
    for k, v in sd.items():
        if v.is_floating_point() and v.numel() > 0 and "running_" not in k:
            t = torch.from_numpy(tensor_for(k, v.shape)).to(v.dtype)
            out[k] = t * gain if (v.dim() > 1 and v.shape[0] > 1) else t
            if norm_weight_offset and v.dim() == 1 and k.endswith("weight"):
                out[k] = out[k] + norm_weight_offset
        else:
            out[k] = v.clone()
    return out
