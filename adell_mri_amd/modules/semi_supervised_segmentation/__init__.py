"""Semi-supervised U-Net of the reference (adell_mri/modules/semi_supervised_segmentation): the
U-Net that also returns its decoder features, the local contrastive loss on them and the training
wrapper that adds that loss (weight 0.01) to the supervised step."""
from .losses import LocalContrastiveLoss  # noqa: F401
from .pl import UNetContrastiveSemiSL  # noqa: F401
from .unet import UNetSemiSL  # noqa: F401
