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
    # the Tversky family (round 4): utils/utils.py:39-59 "tversky_focal" / "combo" / "hybrid_focal" /
    # "unified_focal" of both target families
    "b_focal_alpha": ("binary_focal_loss", dict(gamma=2.0, alpha=0.3, scale=1.5), "binary"),
    "b_tversky": ("binary_focal_tversky_loss", dict(alpha=0.3, beta=0.7, gamma=1.5), "binary"),
    "b_combo": ("combo_loss", dict(alpha=0.4, weight=2.0, gamma=2.0, scale=1.5), "binary"),
    "b_hybrid": ("hybrid_focal_loss",
                 dict(lam=0.3, focal_params=dict(gamma=2.0, alpha=None),
                      tversky_params=dict(alpha=0.3, beta=0.7, gamma=1.3)), "binary"),
    "b_unified": ("unified_focal_loss", dict(weight=0.6, gamma=0.7, lam=0.4), "binary"),
    "mc_tversky": ("mc_focal_tversky_loss",
                   dict(alpha=[0.3, 0.5, 0.7], beta=[0.7, 0.5, 0.3], gamma=1.5), "onehot"),
    "mc_combo": ("mc_combo_loss", dict(alpha=0.4, weight=[1.0, 2.0, 0.5], scale=1.0), "onehot"),
    "mc_hybrid": ("mc_hybrid_focal_loss",
                  dict(lam=0.3, focal_params=dict(alpha=None, gamma=2.0),
                       tversky_params=dict(alpha=0.3, beta=0.7, gamma=1.3)), "index"),
    "mc_unified": ("mc_unified_focal_loss", dict(delta=0.6, gamma=0.7, lam=0.4), "onehot"),
}
