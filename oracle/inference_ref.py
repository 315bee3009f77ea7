"""TEST INFRASTRUCTURE ONLY (see oracle/README): plain restatement of the reference's whole-volume
inference arithmetic, used to check `adell_mri_amd/utils/inference.py`.

Follows `adell_mri/utils/inference.py`: window enumeration and edge adjustment :439-457 and
:633-668, sum / denominator accumulation :690-724, final division :783, flip averaging :382-391.
PINNED (round 3): tests/golden/inference_ops.npz holds outputs of the REAL reference operators
(oracle/make_golden_inference.py: the module imports with a stub `monai.data.meta_tensor`, its only
MONAI use being `isinstance(x, MetaTensor)`); tests/test_inference.py checks this restatement and the
mirror against them. The reference's own tests (`testing/test_segmentation_inference_pl.py:21-52`:
identity round trip, shapes, value range) are repeated there on both implementations as well.
"""
import numpy as np


def windows_3d(shape, window, stride):
    out = []
    for i in range(0, shape[0], stride[0]):
        for j in range(0, shape[1], stride[1]):
            for k in range(0, shape[2], stride[2]):
                lo = [i, j, k]
                hi = [i + window[0], j + window[1], k + window[2]]
                for a in range(3):
                    if hi[a] > shape[a]:
                        lo[a], hi[a] = shape[a] - window[a], shape[a]
                out.append(tuple((lo[a], hi[a]) for a in range(3)))
    return out


def sliding_window_3d(x, fn, window, stride, n_classes):
    """x: numpy [B, C, X, Y, Z]; fn: numpy [B, C, w, w, w] -> numpy [B, n_classes, w, w, w]."""
    B = x.shape[0]
    total = np.zeros((B, n_classes) + x.shape[2:], dtype=np.float64)
    count = np.zeros_like(total)
    for (a1, a2), (b1, b2), (c1, c2) in windows_3d(x.shape[2:], window, stride):
        pred = fn(x[:, :, a1:a2, b1:b2, c1:c2])
        total[:, :, a1:a2, b1:b2, c1:c2] += pred
        count[:, :, a1:a2, b1:b2, c1:c2] += 1.0
    return total / count


def flipped(x, fn, flips):
    out = np.array(fn(x), dtype=np.float64)
    for axes in flips:
        out = out + np.flip(fn(np.flip(x, axes)), axes)
    return out / (len(flips) + 1)
