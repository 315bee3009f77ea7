"""Segmentation losses on the HIP path (mirror of the pieces of
adell_mri/modules/segmentation/losses.py the U-Net training step uses).

``binary_generalized_dice_loss`` / ``binary_focal_loss`` keep the reference's
names and argument names; ``CompoundLoss`` keeps its constructor and
list-returning ``forward`` (losses.py:811-885). When a CompoundLoss holds exactly
the dice + focal pair (the YAML of BASELINE config 2) both losses come out of
ONE pass over the volume (``DiceFocalFn``).
"""
import torch

from ... import ops

eps = 1e-6


class DiceFocalFn(torch.autograd.Function):
    """(dice[B], focal[B]) of probabilities ``pred`` against ``target``."""

    @staticmethod
    def forward(ctx, pred, target, smooth, dice_eps, gamma, focal_eps):
        pred, target = pred.contiguous(), target.contiguous().to(torch.float32)
        dice, focal, sums = ops.dice_focal_fwd(pred, target, smooth, dice_eps, gamma, focal_eps)
        ctx.save_for_backward(pred, target, sums)
        ctx.conf = (smooth, dice_eps, gamma, focal_eps)
        return dice, focal

    @staticmethod
    def backward(ctx, gdice, gfocal):
        pred, target, sums = ctx.saved_tensors
        smooth, dice_eps, gamma, focal_eps = ctx.conf
        # per-item upstream gradients stay on the device (no host read-back / sync)
        dp = ops.dice_focal_bwd_dev(pred, target, sums, smooth, dice_eps, gamma, focal_eps,
                                    gdice, gfocal)
        return dp, None, None, None, None, None


def _check_binary(pred, target, **unsupported):
    if pred.shape != target.shape:
        raise NotImplementedError("class-index targets are outside the HIP path built so far")
    for k, (v, default) in unsupported.items():
        if v != default:
            raise NotImplementedError(f"{k}={v!r} is outside the HIP path built so far")


def binary_generalized_dice_loss(pred, target, weight: float = 1.0, smooth: float = 1.0,
                                 scale: float = 1.0, eps: float = eps) -> torch.Tensor:
    _check_binary(pred, target, weight=(weight, 1.0), scale=(scale, 1.0))
    return DiceFocalFn.apply(pred, target, float(smooth), float(eps), 1.0, 1e-6)[0]


def binary_focal_loss(pred, target, gamma: float, alpha: float = 1.0, threshold: float = 0.5,
                      scale: float = 1.0, label_smoothing: float = 0.0,
                      eps: float = eps) -> torch.Tensor:
    _check_binary(pred, target, alpha=(alpha, 1.0), threshold=(threshold, 0.5),
                  scale=(scale, 1.0), label_smoothing=(label_smoothing, 0.0))
    return DiceFocalFn.apply(pred, target, 0.0, 1e-6, float(gamma), float(eps))[1]


class CompoundLoss(torch.nn.Module):
    def __init__(self, loss_fns_and_kwargs: list, loss_weights: list = None):
        super().__init__()
        self.loss_fns_and_kwargs = [(fn, {} if kw is None else kw)
                                    for fn, kw in loss_fns_and_kwargs]
        self.loss_weights = loss_weights
        if self.loss_weights is None:
            self.loss_weights = [1.0 for _ in self.loss_fns_and_kwargs]
        if len(self.loss_weights) != len(self.loss_fns_and_kwargs):
            raise Exception("loss_weights and loss_fns_and_kwargs should have same length")

    def __setitem__(self, key, value):
        for _, kw in self.loss_fns_and_kwargs:
            kw[key] = value

    def replace_item(self, key, value):
        for _, kw in self.loss_fns_and_kwargs:
            if key in kw:
                kw[key] = value

    def convert_args(self, fn: callable):
        self.loss_fns_and_kwargs = [(f, fn(kw)) for f, kw in self.loss_fns_and_kwargs]

    def _fused_pair(self):
        fns = [f for f, _ in self.loss_fns_and_kwargs]
        if fns == [binary_generalized_dice_loss, binary_focal_loss]:
            dk, fk = (kw for _, kw in self.loss_fns_and_kwargs)
            if set(dk) <= {"smooth", "eps"} and set(fk) <= {"gamma", "eps"} and "gamma" in fk:
                return (float(dk.get("smooth", 1.0)), float(dk.get("eps", eps)),
                        float(fk["gamma"]), float(fk.get("eps", eps)))
        return None

    def forward(self, pred: torch.Tensor, target: torch.Tensor) -> list:
        fused = self._fused_pair()
        if fused is not None and pred.shape == target.shape:
            dice, focal = DiceFocalFn.apply(pred, target, *fused)
            return [dice * self.loss_weights[0], focal * self.loss_weights[1]]
        out = []
        for (loss_fn, kwargs), w in zip(self.loss_fns_and_kwargs, self.loss_weights):
            out.append(loss_fn(pred, target, **kwargs) * w)
        return out
