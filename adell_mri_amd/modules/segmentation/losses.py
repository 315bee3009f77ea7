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
    def forward(ctx, pred, target, smooth, dice_eps, gamma, focal_eps, focal_alpha=1.0):
        pred, target = pred.contiguous(), target.contiguous().to(torch.float32)
        dice, focal, sums = ops.dice_focal_fwd(pred, target, smooth, dice_eps, gamma, focal_eps,
                                               focal_alpha)
        ctx.save_for_backward(pred, target, sums)
        ctx.conf = (smooth, dice_eps, gamma, focal_eps, focal_alpha)
        return dice, focal

    @staticmethod
    def backward(ctx, gdice, gfocal):
        pred, target, sums = ctx.saved_tensors
        smooth, dice_eps, gamma, focal_eps, focal_alpha = ctx.conf
        # per-item upstream gradients stay on the device (no host read-back / sync)
        dp = ops.dice_focal_bwd_dev(pred, target, sums, smooth, dice_eps, gamma, focal_eps,
                                    gdice, gfocal, focal_alpha)
        return dp, None, None, None, None, None, None


def _check_binary(pred, target, **unsupported):
    if pred.shape != target.shape:
        raise NotImplementedError("class-index targets are outside the HIP path built so far")
    for k, (v, default) in unsupported.items():
        if v != default:
            raise NotImplementedError(f"{k}={v!r} is outside the HIP path built so far")


def _scalar(v):
    return float(v.reshape(-1)[0]) if torch.is_tensor(v) else float(v)


def binary_generalized_dice_loss(pred, target, weight: float = 1.0, smooth: float = 1.0,
                                 scale: float = 1.0, eps: float = eps) -> torch.Tensor:
    # (a scalar `weight` multiplies numerator and denominator of generalised_dice_score alike,
    # losses.py:14-54: it cancels, so any value gives the weight-1 result)
    _check_binary(pred, target, scale=(scale, 1.0))
    if torch.is_tensor(weight) and weight.numel() > 1:
        raise NotImplementedError("binary dice: a per-class weight vector is outside the HIP path")
    return DiceFocalFn.apply(pred, target, float(smooth), float(eps), 1.0, 1e-6)[0]


def binary_focal_loss(pred, target, gamma: float, alpha: float = 1.0, threshold: float = 0.5,
                      scale: float = 1.0, label_smoothing: float = 0.0,
                      eps: float = eps) -> torch.Tensor:
    """losses.py:112-164: -mean((alpha p^gamma log p) t + (q^gamma log q) (1 - t)) * scale."""
    _check_binary(pred, target, threshold=(threshold, 0.5), label_smoothing=(label_smoothing, 0.0))
    out = DiceFocalFn.apply(pred, target, 0.0, 1e-6, _scalar(gamma), float(eps), _scalar(alpha))[1]
    return out if float(scale) == 1.0 else out * float(scale)


def _focal_b1(*args, **kwargs):
    """binary_focal_loss in the reference's own output shape [B, 1] (losses.py:152-163 flattens
    from dim 2 and averages the last axis): the composite losses add it to [B] terms, which
    broadcasts to [B, B] there -- reproduced, so that every reduction a caller applies (CompoundLoss
    takes the mean, losses.py:862-885) sees the same numbers."""
    return binary_focal_loss(*args, **kwargs).reshape(-1, 1)


class _ClassSumsFn(torch.autograd.Function):
    """sums[B, C, 3] = (sum p t, sum p, sum t) over the voxels of [B, V, C] tensors
    (ops.class_sums_fwd); differentiable in p."""

    @staticmethod
    def forward(ctx, p, t):
        p, t = p.contiguous(), t.contiguous().to(torch.float32)
        ctx.save_for_backward(t)
        return ops.class_sums_fwd(p, t)

    @staticmethod
    def backward(ctx, gsums):
        (t,) = ctx.saved_tensors
        return ops.class_sums_bwd(t, gsums), None


def _tversky_terms(pred, target):
    """(tp, "fn", "fp") as the reference names them (losses.py:325-330, 690-696):
    sum p t, sum p (1 - t), sum (1 - p) t per (item, class)."""
    s = _ClassSumsFn.apply(_as_bvc(pred), _as_bvc(target))
    tp = s[..., 0]
    return tp, s[..., 1] - tp, s[..., 2] - tp


def binary_focal_tversky_loss(pred, target, alpha: float, beta: float,
                              gamma: float = 1) -> torch.Tensor:
    """losses.py:295-337: 1 - ((tp + 1) / (tp + alpha fn + beta fp + 1))^gamma per item."""
    if pred.shape != target.shape:
        raise ValueError("binary_focal_tversky_loss: pred and target shapes differ")
    B = pred.shape[0]
    tp, fn, fp = _tversky_terms(pred.reshape(B, 1, -1), target.reshape(B, 1, -1))
    nd = (tp + 1) / (tp + _scalar(alpha) * fn + _scalar(beta) * fp + 1)
    return (1 - nd ** _scalar(gamma)).reshape(B)


def combo_loss(pred, target, alpha: float = 0.5, weight: float = 1, gamma: float = 1.0,
               scale: float = 1.0, eps: float = eps) -> torch.Tensor:
    """losses.py:339-383 (as written there: the dice term is called positionally, so `eps` lands
    in its `smooth` argument)."""
    bdl = binary_generalized_dice_loss(pred, target, weight, eps) * scale
    bce = _focal_b1(pred=pred, target=target, alpha=weight, gamma=gamma, scale=scale)
    return _scalar(alpha) * bce + (1 - _scalar(alpha)) * bdl


def hybrid_focal_loss(pred, target, lam: float = 0.5, focal_params: dict = {},
                      tversky_params: dict = {}) -> torch.Tensor:
    """losses.py:386-418: lam * focal + (1 - lam) * focal Tversky (a numeric / None focal alpha
    is replaced by 1, as the reference does)."""
    focal_params = dict(focal_params)
    a = focal_params.get("alpha")
    if a is None or isinstance(a, (int, float)):
        focal_params["alpha"] = 1.0
    bfl = _focal_b1(pred, target, **focal_params)
    bftl = binary_focal_tversky_loss(pred, target, **tversky_params)
    return lam * bfl + (1 - lam) * bftl


def unified_focal_loss(pred, target, weight: float, gamma: float, lam: float = 0.5,
                       threshold: float = 0.5, scale: float = 1.0) -> torch.Tensor:
    """losses.py:421-461, argument for argument: the focal term is called positionally as
    binary_focal_loss(pred, target, weight, 1 - gamma, threshold, scale), i.e. its `gamma` is
    `weight` and its `alpha` is `1 - gamma`."""
    w, g = _scalar(weight), _scalar(gamma)
    bfl = _focal_b1(pred, target, w, 1 - g, threshold, scale)
    bftl = binary_focal_tversky_loss(pred, target, w, 1 - w, g)
    return lam * bfl + (1 - lam) * bftl


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


def mc_focal_tversky_loss(pred, target, alpha, beta, gamma=1.0) -> torch.Tensor:
    """losses.py:656-698: n = tp + 1, d = n + alpha fn + beta fp + 1 (the second + 1 as written
    there), mean over the classes of 1 - (n / d)^gamma."""
    target = _mc_target(pred, target)
    tp, fn, fp = _tversky_terms(pred, target)
    C = pred.shape[1]
    n = tp + 1
    d = n + _class_vector(alpha, C, pred) * fn + _class_vector(beta, C, pred) * fp + 1
    g = _class_vector(gamma, C, pred)
    return torch.mean(1 - torch.pow(n / d, g), dim=-1)


def mc_combo_loss(pred, target, alpha: float = 0.5, weight=1, scale: float = 1.0) -> torch.Tensor:
    """losses.py:701-734 (positional calls as written: `scale` lands in the dice loss's `smooth`)."""
    bdl = mc_generalized_dice_loss(pred, target, weight, scale)
    bce = cat_cross_entropy(pred, target, weight, scale)
    return _scalar(alpha) * bce + (1 - _scalar(alpha)) * bdl


def mc_hybrid_focal_loss(pred, target, lam: float = 1.0, focal_params: dict = {},
                         tversky_params: dict = {}) -> torch.Tensor:
    """losses.py:737-769."""
    focal_params = dict(focal_params)
    a = focal_params.get("alpha")
    if a is None or isinstance(a, (int, float)):
        focal_params["alpha"] = 1.0
    fl = mc_focal_loss(pred, target, **focal_params)
    ftl = mc_focal_tversky_loss(pred, target, **tversky_params)
    return lam * fl + (1 - lam) * ftl


def mc_unified_focal_loss(pred, target, delta, gamma, lam: float, scale: float = 1.0) -> torch.Tensor:
    """losses.py:772-808: mc_focal_loss(pred, target, delta, 1 - gamma, scale) and
    mc_focal_tversky_loss(pred, target, delta, 1 - delta, gamma)."""
    C = pred.shape[1]
    dvec = _class_vector(delta, C, pred)
    fl = mc_focal_loss(pred, target, dvec, 1 - _scalar(gamma), scale)
    ftl = mc_focal_tversky_loss(pred, target, dvec, 1 - dvec, gamma)
    return lam * fl + (1 - lam) * ftl


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
