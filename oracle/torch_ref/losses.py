"""CPU restatement (stock torch) of the element-wise segmentation losses of the reference
(TEST INFRASTRUCTURE ONLY; adell_mri/modules/segmentation/losses.py:14-54, 79-109, 528-653),
pinned to tests/golden/losses_mc.npz by tests/test_oracle_golden.py."""
import torch


def _cw(w, pred):
    w = torch.as_tensor(w, dtype=pred.dtype).flatten()
    shape = [1] * pred.dim()
    shape[1] = w.numel()
    return w.reshape(shape)


def binary_cross_entropy(pred, target, weight=1.0, scale=1.0, label_smoothing=0.0, eps=1e-6):
    t = (target * (1 - label_smoothing) + label_smoothing / 2).flatten(1)
    p = pred.flatten(1)
    return -((weight * t * torch.log(p + eps) + (1 - t) * torch.log(1 - p + eps)) * scale).mean(1)


def one_hot3(x):
    nd = x.dim()
    return torch.nn.functional.one_hot(x.long(), 3).permute(0, nd, *range(1, nd))


def cat_cross_entropy(pred, target, weight=1.0, scale=1.0, label_smoothing=0.0, eps=1e-6):
    if pred.shape != target.shape:
        target = one_hot3(target)
    t = target * (1 - label_smoothing) + 1 / target.shape[1]
    return ((-t * torch.log(pred + eps)) * _cw(weight, pred)).flatten(1).mul(scale).mean(1)


def mc_focal_loss(pred, target, alpha, gamma, scale=1.0, label_smoothing=0.0, eps=1e-6):
    if pred.shape != target.shape:
        target = one_hot3(target)
    pt = torch.where(target > 0.5, pred, 1 - pred)
    t = target * (1 - label_smoothing) + 1 / target.shape[1]
    ce = -t * torch.log(pred + eps)
    return (_cw(alpha, pred) * (1 - pt + eps) ** gamma * ce).flatten(1).mul(scale).mean(1)


def mc_generalized_dice_loss(pred, target, weight=1.0, smooth=1.0, scale=1.0, eps=1e-6):
    if pred.shape != target.shape:
        target = one_hot3(target)
    w = torch.as_tensor(weight, dtype=pred.dtype).flatten()[None]
    t, p = target.flatten(2).to(pred.dtype), pred.flatten(2)
    num = (w * torch.clip(t * p * scale, 0).sum(-1)).sum(-1)
    den = (w * torch.clip((t + p + smooth) * scale, eps).sum(-1)).sum(-1)
    return 1 - 2 * num / den
