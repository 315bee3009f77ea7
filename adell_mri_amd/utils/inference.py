"""Whole-volume inference around the forward kernels: sliding windows, test-time flips, MC
dropout (reference `adell_mri/utils/inference.py`).

Mirrors of `TensorListReduction` (:262-300), `FlippedInference` (:303-391),
`SlidingWindowSegmentation` (:394-784) and `SegmentationInference` (:787-990): same constructor
arguments and call semantics, for torch tensors or one-level dicts / lists / tuples of tensors
(numpy inputs and MONAI MetaTensor metadata are data-pipeline concerns and out of scope: a numpy
array raises). The window plan is computed once per shape; windows are cropped, batched and
accumulated on the device the input lives on, and the network call is whatever
`inference_function` is -- for this package `UNetPL.predict_step` / `UNet.forward` on the HIP
kernels."""
from itertools import product
from typing import Callable, List, Sequence

import torch


def _example(X) -> torch.Tensor:
    if isinstance(X, torch.Tensor):
        return X
    if isinstance(X, dict):
        return X[next(iter(X))]
    if isinstance(X, (tuple, list)):
        return X[0]
    raise NotImplementedError("Supported inputs are torch.Tensor, dict, tuple, list")


def _map(X, fn, keys=None):
    """Applies fn to the tensors of a possibly one-level nested structure (dict entries / list
    positions not in `keys`, when keys is given, are passed through)."""
    if isinstance(X, torch.Tensor):
        return fn(X)
    if isinstance(X, dict):
        return {k: (fn(v) if (keys is None or k in keys) and isinstance(v, torch.Tensor) else v)
                for k, v in X.items()}
    if isinstance(X, (tuple, list)):
        return [fn(v) if (keys is None or i in keys) and isinstance(v, torch.Tensor) else v
                for i, v in enumerate(X)]
    raise NotImplementedError("Supported inputs are torch.Tensor, dict, tuple, list")


def _collate(items: list, batched: bool):
    """A list of structures -> one structure of batches: cat along dim 0 when the windows
    already carry a batch dimension, stack otherwise (reference :232-259)."""
    join = torch.cat if batched else torch.stack
    first = items[0]
    if isinstance(first, torch.Tensor):
        return join(items, 0)
    if isinstance(first, dict):
        return {k: join([it[k] for it in items], 0) for k in first}
    return [join([it[i] for it in items], 0) for i in range(len(first))]


def window_plan(shape: Sequence[int], window: Sequence[int], stride: Sequence[int]):
    """Window bounds ((a1, a2), (b1, b2)[, (c1, c2)]) in the reference's order: starts at
    multiples of the stride, a window that would stick out is moved back so that it ends at the
    edge (reference :439-457, :601-668)."""
    axes = []
    for size, w, s in zip(shape, window, stride):
        bounds = []
        for start in range(0, size, s):
            lo, hi = start, start + w
            if hi > size:
                lo, hi = size - w, size
            bounds.append((lo, hi))
        axes.append(bounds)
    return list(product(*axes))


class TensorListReduction:
    """Mean of a list of tensors with optional pre- / post-processing (reference :262-300)."""

    def __init__(self, preproc_fn: Callable = None, postproc_fn: Callable = None,
                 strategy: str = "mean"):
        assert strategy in ["mean"]
        self.preproc_fn = preproc_fn
        self.postproc_fn = postproc_fn
        self.strategy = strategy

    def __call__(self, X):
        if isinstance(X, (list, tuple)):
            if self.preproc_fn is not None:
                X = [self.preproc_fn(x) for x in X]
            X = torch.stack(list(X)).mean(0)
        elif self.preproc_fn is not None:
            X = self.preproc_fn(X)
        if self.postproc_fn is not None:
            X = self.postproc_fn(X)
        return X


class FlippedInference:
    """Mean of the prediction and of the un-flipped predictions of flipped inputs
    (reference :303-391; `flips` are lists of absolute tensor dimensions)."""

    def __init__(self, inference_function: Callable, flips: List[List[int]],
                 flip_keys: List[str] = None, ndim: int = 3, inference_batch_size: int = 1):
        self.inference_function = inference_function
        self.flips = flips
        self.flip_keys = flip_keys
        self.ndim = ndim

    def flip(self, X, axis):
        axis = tuple(axis)
        return _map(X, lambda t: torch.flip(t, axis), self.flip_keys)

    def __call__(self, X, *args, **kwargs):
        output = self.inference_function(X, *args, **kwargs).clone()
        for flip in self.flips:
            pred = self.inference_function(self.flip(X, flip), *args, **kwargs)
            output += torch.flip(pred, tuple(flip))
        return output / (len(self.flips) + 1)


class SlidingWindowSegmentation:
    """Runs `inference_function` on every window of the input and averages the predictions where
    windows overlap (reference :394-784). Inputs are batched with a channel dimension
    ([B, C, *spatial]) or unbatched ([C, *spatial]); structures hold tensors of one spatial size."""

    def __init__(self, sliding_window_size: Sequence[int], inference_function: Callable,
                 n_classes: int, stride: Sequence[int] = None, inference_batch_size: int = 1):
        self.sliding_window_size = sliding_window_size
        self.inference_function = inference_function
        self.n_classes = n_classes
        self.stride = stride if stride is not None else sliding_window_size
        self.inference_batch_size = inference_batch_size
        self.ndim = len(sliding_window_size)

    def get_all_crops(self, X):
        """Yields (cropped structure, coords) for every window."""
        for coords in window_plan(_example(X).shape[-self.ndim:], self.sliding_window_size,
                                  self.stride):
            yield self.extract_patch(X, coords), coords

    def extract_patch(self, X, coords):
        """The window of every tensor of the structure; entries that are not tensors (file names,
        metadata) are dropped, as the reference does (:562-599)."""
        index = (Ellipsis,) + tuple(slice(lo, hi) for lo, hi in coords)
        if isinstance(X, dict):
            return {k: v[index] for k, v in X.items() if isinstance(v, torch.Tensor)}
        if isinstance(X, (tuple, list)):
            return [v[index] for v in X if isinstance(v, torch.Tensor)]
        return _map(X, lambda t: t[index])

    def __call__(self, X, *args, **kwargs):
        example = _example(X)
        if not isinstance(example, torch.Tensor):
            raise NotImplementedError("SlidingWindowSegmentation: torch tensors only")
        out_size = list(example.shape)
        if len(out_size) == self.ndim + 2:
            batched = True
            out_size[1] = self.n_classes
        elif len(out_size) < self.ndim + 2:
            batched = False
            out_size[0] = self.n_classes
        else:
            raise Exception("length of input array shape should be <= self.ndim+2")
        per_window = out_size[0]   # rows of the prediction that belong to one window (:762)
        total = torch.zeros(out_size, device=example.device)
        count = torch.zeros(out_size, device=example.device)
        pending, pending_coords = [], []

        def flush():
            with torch.no_grad():
                pred = self.inference_function(_collate(pending, batched), *args, **kwargs)
            for out, coords in zip(torch.split(pred, per_window, 0), pending_coords):
                index = (Ellipsis,) + tuple(slice(lo, hi) for lo, hi in coords)
                total[index] += out.squeeze(0).squeeze(0)
                count[index] += 1.0
            pending.clear()
            pending_coords.clear()

        for crop, coords in self.get_all_crops(X):
            pending.append(crop)
            pending_coords.append(coords)
            if len(pending) == self.inference_batch_size:
                flush()
        if pending:
            flush()
        return total / count


class SegmentationInference:
    """Sliding windows + flips + MC dropout + reduction over several networks in one callable
    (reference :787-990)."""

    def __init__(self, base_inference_function, sliding_window_size: List[int] = None,
                 stride=None, inference_batch_size: int = 1, n_classes: int = 2,
                 flip: bool = False, flip_keys: List[str] = ["image"], mc_iterations: int = None,
                 ndim: int = 3, reduction: Callable = None):
        self.base_inference_function = base_inference_function
        self.sliding_window_size = sliding_window_size
        self.n_classes = n_classes
        self.stride = stride
        self.flip = flip
        self.flip_keys = flip_keys
        self.mc_iterations = mc_iterations
        self.ndim = ndim
        self.inference_batch_size = inference_batch_size
        self.reduction = reduction
        self.flips = [(2,)]
        self.update_base_inference_function(base_inference_function)

    def _wrap(self, fn):
        if self.sliding_window_size is not None:
            fn = SlidingWindowSegmentation(
                sliding_window_size=self.sliding_window_size, inference_function=fn,
                n_classes=self.n_classes if self.n_classes > 2 else 1, stride=self.stride,
                inference_batch_size=self.inference_batch_size)
        if self.flip is True:
            fn = FlippedInference(inference_function=fn, flips=self.flips,
                                  flip_keys=self.flip_keys, ndim=self.ndim,
                                  inference_batch_size=self.inference_batch_size)
        return fn

    def update_base_inference_function(self, base_inference_function):
        if base_inference_function is None:
            return
        if self.sliding_window_size is not None and isinstance(self.stride, float):
            self.stride = [int(x * self.stride) for x in self.sliding_window_size]
        if isinstance(base_inference_function, (list, tuple)):
            self.inference_function = [self._wrap(fn) for fn in base_inference_function]
        else:
            self.inference_function = self._wrap(base_inference_function)

    def call_regular(self, X, *args, **kwargs):
        if isinstance(self.inference_function, (list, tuple)):
            output = [fn(X, *args, **kwargs) for fn in self.inference_function]
        else:
            output = self.inference_function(X, *args, **kwargs)
        if self.reduction is not None:
            output = self.reduction(output)
        return output

    def call_dropout(self, X, *args, **kwargs):
        outputs = torch.stack([self.call_regular(X, *args, **kwargs)
                               for _ in range(self.mc_iterations)])
        return torch.cat([outputs.mean(0), outputs.std(0)], dim=1)

    def __call__(self, X, *args, **kwargs):
        with torch.no_grad():
            if self.mc_iterations is not None:
                return self.call_dropout(X, *args, **kwargs)
            return self.call_regular(X, *args, **kwargs)
