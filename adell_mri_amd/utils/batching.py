"""The step before the hot path: batch assembly.

* `unpack_crops`, `safe_collate`, `safe_collate_crops`: mirrors of the collate functions the
  reference hands to its DataLoaders (`adell_mri/utils/utils.py:230-244, 308-377`): one level deep,
  stacking where shapes allow and returning the list otherwise.
* `DeviceCropSampler`: the MI355X-side replacement for the random-crop / random-flip part of the
  MONAI transform chain (`transform_factory/augmentations.py:19-178`, `train.py:359-364`): volumes
  stay resident in HBM (288 GB hold a whole training set of 256x256x128 studies) and every crop of
  a batch is cut, flipped and laid out NDHWC by one gather launch per tensor -- no worker
  processes, no host round trip. It is not an API mirror (MONAI is out of scope); the crop
  arithmetic is checked against plain slicing and `torch.flip`."""
from typing import Dict, List, Sequence

import torch

from .. import ops


def unpack_crops(X: list) -> list:
    """[[a, b], [c]] -> [a, b, c] (utils.py:230-244)."""
    return [xx for x in X for xx in x]


def _stack_or_list(elements):
    try:
        elements = [torch.as_tensor(y) for y in elements]
    except Exception:
        return elements
    try:
        return torch.stack(elements)
    except Exception:
        return elements


def safe_collate(X: list):
    """Default-collate one level deep; incompatible shapes stay a list (utils.py:308-359)."""
    example = X[0]
    if isinstance(example, list):
        return [_stack_or_list(elements) for elements in zip(*X)]
    if isinstance(example, dict):
        return {k: _stack_or_list([x[k] if k in x else None for x in X]) for k in example}
    return None


def safe_collate_crops(X: list):
    """`safe_collate` over the flattened output of a cropping transform (utils.py:362-377)."""
    return safe_collate(unpack_crops(X))


class DeviceCropSampler:
    """Random crops with random flips of device-resident [C, X, Y, Z] (or [C, X, Y]) volumes.

    ``__call__(volumes, n)`` takes a dict of tensors of one spatial size and returns a dict of
    [n, C, *crop_size] batches (channels-last memory, as the kernels want them); all tensors of a
    sample share the crop origin and the flips. ``plan(shape, n)`` exposes the drawn
    (origin, flips) list so that a caller can reproduce or log it."""

    def __init__(self, crop_size: Sequence[int], flip_axes: Sequence[int] = (),
                 flip_prob: float = 0.5, seed: int = 42):
        self.crop_size = [int(c) for c in crop_size]
        self.flip_axes = tuple(int(a) for a in flip_axes)
        self.flip_prob = float(flip_prob)
        self.generator = torch.Generator().manual_seed(int(seed))

    def plan(self, shape: Sequence[int], n: int) -> List[tuple]:
        out = []
        for _ in range(n):
            origin = [int(torch.randint(0, s - c + 1, (1,), generator=self.generator))
                      for s, c in zip(shape, self.crop_size)]
            flips = tuple(a for a in self.flip_axes
                          if float(torch.rand((1,), generator=self.generator)) < self.flip_prob)
            out.append((origin, flips))
        return out

    def cut(self, vol: torch.Tensor, plan: List[tuple]) -> torch.Tensor:
        """One tensor -> [len(plan), C, *crop] batch; one gather launch per crop."""
        if not vol.is_cuda or vol.dtype != torch.float32:
            raise ops.AdellHipError("DeviceCropSampler: float32 CUDA tensors only")
        vol = vol.contiguous()
        nd = vol.dim() - 1
        C, shape = vol.shape[0], list(vol.shape[1:])
        crop = self.crop_size
        strides = [1] * nd
        for a in range(nd - 2, -1, -1):
            strides[a] = strides[a + 1] * shape[a + 1]
        cstride = strides[0] * shape[0]
        n = len(plan)
        full = crop if nd == 3 else [1] + crop
        batch = ops.new_act(n, C, *full, vol.device)      # [n, C, D, H, W] logical, NDHWC memory
        flat = batch.permute(0, 2, 3, 4, 1).reshape(n, -1)
        for i, (origin, flips) in enumerate(plan):
            dims, axes = [], []
            for a in range(nd):
                if a in flips:   # index = origin + size - 1 - c
                    dims.append((crop[a], a, -1))
                    axes.append((shape[a], strides[a], origin[a] + crop[a] - 1))
                else:
                    dims.append((crop[a], a, 1))
                    axes.append((shape[a], strides[a], origin[a]))
            dims.append((C, nd, 1))
            axes.append((C, cstride, 0))
            ops.gather_nd(vol, dims, axes, out=flat[i])
        return batch if nd == 3 else batch.squeeze(2)

    def __call__(self, volumes: Dict[str, torch.Tensor], n: int) -> Dict[str, torch.Tensor]:
        first = next(iter(volumes.values()))
        plan = self.plan(list(first.shape[1:]), n)
        return {k: self.cut(v, plan) for k, v in volumes.items()}
