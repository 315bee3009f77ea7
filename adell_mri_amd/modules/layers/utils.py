"""Small tensor helpers (mirror of adell_mri/modules/layers/utils.py)."""
from typing import List

import torch


def split_int_into_n(i: int, n: int) -> List[int]:
    """``i`` divided into ``n`` slots as equally as possible, the remainder to the first slots
    (adell_mri/modules/layers/utils.py:8-27)."""
    out = [i // n] * n
    for idx in range(i % n):
        out[idx] += 1
    return out


def crop_to_size(X: torch.Tensor, output_size: List[int]) -> torch.Tensor:
    """Centre-crop the spatial dims of ``X`` ([N, C, ...]) to ``output_size``.

    Same window as adell_mri/modules/layers/utils.py:30-52 (offset ``diff // 2``),
    expressed as a strided view instead of ``index_select`` copies.
    """
    rows = getattr(X, "_adell_rows", None)
    if (X.dim() == 5 and X.is_cuda and rows is None and X.dtype == torch.float32
            and tuple(X.shape[2:]) != tuple(output_size)):
        from ... import functional as HF

        return HF.crop3d(X, output_size)        # one kernel each way instead of a strided view
    sl = [slice(None), slice(None)]
    for cur, out in zip(X.shape[2:], output_size):
        a = (cur - out) // 2
        sl.append(slice(a, a + out))
    out = X[tuple(sl)]
    # a spatial crop of a split-row tensor (functional.expect_rows) keeps whole voxels, i.e. whole
    # rows: still a split-row tensor
    if rows is not None and out is not X:
        out._adell_rows = rows
    return out
