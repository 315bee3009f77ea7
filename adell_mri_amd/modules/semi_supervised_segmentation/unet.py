"""``UNetSemiSL`` (mirror of adell_mri/modules/semi_supervised_segmentation/unet.py:13-223): the
U-Net plus a 1x1 ``linear_transformation`` on its last decoder features and ``forward_features``.
``forward`` itself (with ``return_features`` / ``return_bottleneck`` / ``return_logits``) is the
one ``UNet`` already has; every layer runs on the HIP kernels."""
import torch

from ..layers.conv import Conv2d, Conv3d
from ..segmentation.unet import UNet


class UNetSemiSL(UNet):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.init_linear_transformation()

    def init_linear_transformation(self, *args, **kwargs):
        """1x1 convolution depth[0] -> depth[0] applied to the teacher's features
        (semi_supervised_segmentation/unet.py:27-43)."""
        if self.spatial_dimensions == 2:
            self.linear_transformation = Conv2d(self.depth[0], self.depth[0], kernel_size=1)
        elif self.spatial_dimensions == 3:
            self.linear_transformation = Conv3d(self.depth[0], self.depth[0], kernel_size=1)

    def forward_features(self, X: torch.Tensor, X_skip_layer: torch.Tensor = None,
                         X_feature_conditioning: torch.Tensor = None,
                         apply_linear_transformation: bool = False) -> torch.Tensor:
        """Last decoder feature map [B, depth[0], *spatial], optionally through
        ``linear_transformation`` (semi_supervised_segmentation/unet.py:193-223)."""
        encoding_out, bottleneck, X_skip_layer, X_feature_conditioning = self._encode(
            X, X_skip_layer, X_feature_conditioning)
        features, _ = self._run_decoder(encoding_out, bottleneck, X_skip_layer,
                                        X_feature_conditioning)
        if apply_linear_transformation is True:
            features = self.linear_transformation(features)
        return features
