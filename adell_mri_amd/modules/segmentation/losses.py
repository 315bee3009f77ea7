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


class _SegLossFn(torch.autograd.Function):
    """One of the element-wise losses of csrc/loss_optim.hip (ops.SEG_LOSS_KINDS) on
    [B, C, *spatial] probabilities; ``conf`` = (kind, eps, scale, ls, gamma, smooth, w_pos)."""

    @staticmethod
    def forward(ctx, pred, target, cw, conf):
        B, C = pred.shape[:2]
        p3 = _as_bvc(pred)
        t3 = _as_bvc(target.to(torch.float32))
        loss, sums = ops.seg_loss_fwd(conf[0], p3, t3, cw, *conf[1:])
        ctx.save_for_backward(p3, t3, cw, sums)
        ctx.conf, ctx.shape = conf, tuple(pred.shape)
        return loss

    @staticmethod
    def backward(ctx, g):
        p3, t3, cw, sums = ctx.saved_tensors
        dp = ops.seg_loss_bwd(ctx.conf[0], p3, t3, cw, *ctx.conf[1:], sums, g)
        shape = ctx.shape
        B, C = shape[:2]
        dp = dp.view(B, *shape[2:], C)
        nd = len(shape)
        return dp.permute(0, nd - 1, *range(1, nd - 1)), None, None, None


def _as_bvc(x):
    """[B, C, *spatial] (any strides) -> contiguous [B, V, C] (channels-last order)."""
    B, C = x.shape[:2]
    nd = x.dim()
    return x.permute(0, *range(2, nd), 1).reshape(B, -1, C).contiguous()


def _class_vector(w, C, like):
    """Scalar or per-class weights -> float32 [C] on the device of ``like``."""
    w = torch.as_tensor(w, dtype=torch.float32, device=like.device).flatten()
    if w.numel() == 1:
        w = w.expand(C)
    if w.numel() != C:
        raise ValueError(f"expected 1 or {C} class weights, got {w.numel()}")
    return w.contiguous()


def classes_to_one_hot(X: torch.Tensor) -> torch.Tensor:
    """Class-index map -> one-hot with the classes second (losses.py:481-499). As in the
    reference the number of classes is fixed at three."""
    n_dim = X.dim()
    out_dim = [0, n_dim, *range(1, n_dim)]
    return torch.nn.functional.one_hot(X.long(), num_classes=3).permute(out_dim).to(X.device)


def binary_cross_entropy(pred, target, weight: float = 1.0, scale: float = 1.0,
                         label_smoothing: float = 0.0, eps: float = eps) -> torch.Tensor:
    """losses.py:79-109: -mean((w t' log(p + eps) + (1 - t') log(1 - p + eps)) scale) per item,
    t' = t (1 - ls) + ls / 2."""
    if pred.shape != target.shape:
        raise ValueError("binary_cross_entropy: pred and target shapes differ")
    p = pred.reshape(pred.shape[0], 1, -1)
    conf = (ops.SEG_LOSS_KINDS["binary_cross_entropy"], float(eps), float(scale),
            float(label_smoothing), 0.0, 0.0, float(weight))
    return _SegLossFn.apply(p, target.reshape(p.shape), None, conf)


def _mc_target(pred, target):
    if pred.shape != target.shape:
        target = classes_to_one_hot(target)
        if target.shape != pred.shape:
            raise ValueError(f"one-hot target {tuple(target.shape)} does not match the prediction "
                             f"{tuple(pred.shape)} (the reference encodes exactly 3 classes)")
    return target


def cat_cross_entropy(pred, target, weight=1.0, scale: float = 1.0, label_smoothing: float = 0.0,
                      eps: float = eps) -> torch.Tensor:
    """losses.py:528-562 (target' = t (1 - ls) + 1 / C, as written there)."""
    target = _mc_target(pred, target)
    conf = (ops.SEG_LOSS_KINDS["cat_cross_entropy"], float(eps), float(scale),
            float(label_smoothing), 0.0, 0.0, 1.0)
    return _SegLossFn.apply(pred, target, _class_vector(weight, pred.shape[1], pred), conf)


def mc_focal_loss(pred, target, alpha, gamma, scale: float = 1.0, label_smoothing: float = 0.0,
                  eps: float = eps) -> torch.Tensor:
    """losses.py:565-607: alpha[c] (1 - pt + eps)^gamma * ce, mean over classes and voxels."""
    target = _mc_target(pred, target)
    conf = (ops.SEG_LOSS_KINDS["mc_focal"], float(eps), float(scale), float(label_smoothing),
            float(gamma), 0.0, 1.0)
    return _SegLossFn.apply(pred, target, _class_vector(alpha, pred.shape[1], pred), conf)


def mc_generalized_dice_loss(pred, target, weight=1.0, smooth: float = 1.0, scale: float = 1.0,
                             eps: float = eps) -> torch.Tensor:
    """losses.py:610-653: 1 - 2 sum_c w_c num_c / sum_c w_c den_c with the clipped sums of
    generalised_dice_score (:14-54)."""
    target = _mc_target(pred, target)
    conf = (ops.SEG_LOSS_KINDS["mc_dice"], float(eps), float(scale), 0.0, 0.0, float(smooth), 1.0)
    return _SegLossFn.apply(pred, target, _class_vector(weight, pred.shape[1], pred), conf)


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
