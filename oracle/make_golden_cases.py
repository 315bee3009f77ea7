"""Case tables shared by oracle/make_golden.py (needs the reference) and the tests (do not)."""
LOSS_CASES = {
    # name: (function name in segmentation/losses.py, kwargs, target kind)
    "bce": ("binary_cross_entropy", dict(weight=2.0, label_smoothing=0.1, scale=1.5), "binary"),
    "bce_default": ("binary_cross_entropy", {}, "binary"),
    "cat_ce": ("cat_cross_entropy", dict(weight=[1.0, 2.0, 0.5], label_smoothing=0.1), "onehot"),
    "cat_ce_index": ("cat_cross_entropy", dict(weight=1.0), "index"),
    "mc_focal": ("mc_focal_loss", dict(alpha=[1.0, 2.0, 0.5], gamma=2.0), "onehot"),
    "mc_focal_g15": ("mc_focal_loss", dict(alpha=1.0, gamma=1.5, scale=0.5, label_smoothing=0.05),
                     "onehot"),
    "mc_dice": ("mc_generalized_dice_loss", dict(weight=[1.0, 2.0, 0.5], smooth=1e-5), "onehot"),
    "mc_dice_default": ("mc_generalized_dice_loss", {}, "index"),
}
