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


# ---- the Tversky family (losses.py:112-164, 251-292, 295-462, 656-808), round 4 -------------------
def binary_generalized_dice_loss(pred, target, weight=1.0, smooth=1.0, scale=1.0, eps=1e-6):
    w = torch.as_tensor(weight, dtype=pred.dtype)
    t, p = target.flatten(1), pred.flatten(1)
    num = w * torch.clip(t * p * scale, 0).sum(-1)
    den = w * torch.clip((t + p + smooth) * scale, eps).sum(-1)
    return (1 - 2 * num / den).reshape(pred.shape[0])


def binary_focal_loss(pred, target, gamma, alpha=1.0, threshold=0.5, scale=1.0,
                      label_smoothing=0.0, eps=1e-6):
    p = torch.clamp_min(pred, eps).flatten(2)
    q = torch.clamp_min(1 - p, eps)
    t = (target > threshold).to(pred.dtype).flatten(2)
    t = t * (1 - label_smoothing) + label_smoothing / 2
    a = torch.as_tensor(alpha, dtype=pred.dtype)
    # (shape [B, 1] as in the reference: flatten(start_dim=2) keeps the channel axis, so the
    # composite losses below broadcast it against their [B] terms into [B, B])
    return (-(a * p ** gamma * torch.log(p) * t + q ** gamma * torch.log(q) * (1 - t)) * scale).mean(-1)


def binary_focal_tversky_loss(pred, target, alpha, beta, gamma=1):
    p, t = pred.flatten(1), target.flatten(1)
    tp = (p * t).sum(1)
    fn = (p * (1 - t)).sum(1)
    fp = ((1 - p) * t).sum(1)
    return 1 - ((tp + 1) / (tp + alpha * fn + beta * fp + 1)) ** gamma


def combo_loss(pred, target, alpha=0.5, weight=1, gamma=1.0, scale=1.0, eps=1e-6):
    bdl = binary_generalized_dice_loss(pred, target, weight, eps) * scale
    bce = binary_focal_loss(pred, target, alpha=weight, gamma=gamma, scale=scale)
    return alpha * bce + (1 - alpha) * bdl


def hybrid_focal_loss(pred, target, lam=0.5, focal_params={}, tversky_params={}):
    fp = dict(focal_params)
    if fp.get("alpha") is None or isinstance(fp["alpha"], (int, float)):
        fp["alpha"] = 1.0
    return lam * binary_focal_loss(pred, target, **fp) + \
        (1 - lam) * binary_focal_tversky_loss(pred, target, **tversky_params)


def unified_focal_loss(pred, target, weight, gamma, lam=0.5, threshold=0.5, scale=1.0):
    bfl = binary_focal_loss(pred, target, weight, 1 - gamma, threshold, scale)
    bftl = binary_focal_tversky_loss(pred, target, weight, 1 - weight, gamma)
    return lam * bfl + (1 - lam) * bftl


def mc_focal_tversky_loss(pred, target, alpha, beta, gamma=1.0):
    if pred.shape != target.shape:
        target = one_hot3(target)
    p, t = pred.flatten(2), target.flatten(2).to(pred.dtype)
    a = torch.as_tensor(alpha, dtype=pred.dtype)
    b = torch.as_tensor(beta, dtype=pred.dtype)
    g = torch.as_tensor(gamma, dtype=pred.dtype)
    n = (p * t).sum(-1) + 1
    d = n + a * (p * (1 - t)).sum(-1) + b * ((1 - p) * t).sum(-1) + 1
    return torch.mean(1 - torch.pow(n / d, g), dim=-1)


def mc_combo_loss(pred, target, alpha=0.5, weight=1, scale=1.0):
    bdl = mc_generalized_dice_loss(pred, target, weight, scale)
    bce = cat_cross_entropy(pred, target, weight, scale)
    return alpha * bce + (1 - alpha) * bdl


def mc_hybrid_focal_loss(pred, target, lam=1.0, focal_params={}, tversky_params={}):
    fp = dict(focal_params)
    if fp.get("alpha") is None or isinstance(fp["alpha"], (int, float)):
        fp["alpha"] = 1.0
    return lam * mc_focal_loss(pred, target, **fp) + \
        (1 - lam) * mc_focal_tversky_loss(pred, target, **tversky_params)


def mc_unified_focal_loss(pred, target, delta, gamma, lam, scale=1.0):
    d = torch.as_tensor(delta, dtype=pred.dtype)
    fl = mc_focal_loss(pred, target, d, 1 - gamma, scale)
    ftl = mc_focal_tversky_loss(pred, target, d, 1 - d, gamma)
    return lam * fl + (1 - lam) * ftl
