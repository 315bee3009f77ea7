"""Seeded inputs and sample positions of the full-size fixtures (TEST INFRASTRUCTURE ONLY).

Shared by ``oracle/make_golden.py`` (which runs the real reference on them in the build
container) and ``tests/test_fullsize_reference_gpu.py`` (which regenerates the same tensors on
the GPU box: torch's CPU generator stream is identical for the same torch build, and the
fixture stores a checksum of the inputs to prove it)."""
import zlib

import numpy as np
import torch

FULL_SEED = 20260128


def full_inputs(shape, seed=FULL_SEED):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(shape, generator=g)
    y = (torch.rand((shape[0], 1, *shape[2:]), generator=g) > 0.9).float()
    return x, y


def sample_positions(n, count, seed):
    return np.random.default_rng(seed).choice(n, size=min(count, n), replace=False).astype(np.int64)


def zlib_crc(s):
    return zlib.crc32(s.encode())
