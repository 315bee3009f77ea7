"""U-Net++ on the MI355X kernels: drop-in mirror of
``adell_mri.modules.segmentation.unetpp.UNetPlusPlus`` (reference:
adell_mri/modules/segmentation/unetpp.py:17-310): same constructor (:26-48), dense
skip links (``DenseBlock``), auxiliary heads ``final_layer_aux`` and the
``(pred, bn_out, aux)`` return of ``forward`` (:217-310).
"""
from typing import Dict, List

import numpy as np
import torch

from ... import functional as HF
from ..._lib import AdellHipError
from ..layers.standard_blocks import DenseBlock
from ..layers.utils import crop_to_size
from .unet import UNet, _cat_channels, _nearest


class UNetPlusPlus(UNet):
    # signature: adell_mri/modules/segmentation/unetpp.py:26-48 (the U-Net's, without deep
    # supervision / encoder_only; ``link_type`` is accepted and ignored: the links are dense)
    def __init__(self, spatial_dimensions: int = 2,
                 encoding_operations: List[torch.nn.ModuleList] = None,
                 conv_type: str = "regular", link_type: str = "identity",
                 upscale_type: str = "upsample", interpolation: str = "bilinear",
                 norm_type: str = "batch", dropout_type: str = "dropout", padding: int = 0,
                 dropout_param: float = 0.1, activation_fn: torch.nn.Module = torch.nn.PReLU,
                 in_channels: int = 1, n_classes: int = 2, depth: list = [16, 32, 64],
                 kernel_sizes: list = [3, 3, 3], strides: list = [2, 2, 2],
                 bottleneck_classification: bool = False, skip_conditioning: int = None,
                 feature_conditioning: int = None,
                 feature_conditioning_params: Dict[str, torch.Tensor] = None) -> torch.nn.Module:
        passed = {k: v for k, v in locals().items()
                  if k not in ("self", "link_type") and not k.startswith("__")}
        super().__init__(**passed)
        # The reference builds every layer a second time here (unetpp.py:128-145). Because
        # ``encoding_operations`` is no longer None on that second pass it takes the
        # backbone branch, which swaps every strided downsampling conv block for
        # MaxPool(kernel=s, stride=s, padding=s//2): the shipped U-Net++ therefore pools
        # (and relies on crop_to_size in the decoder). Reproduced for checkpoint parity.
        self.init_encoder_backbone()

    def init_link_ops(self):
        ex = self.skip_conditioning if self.skip_conditioning is not None else 0
        self.link_ops = torch.nn.ModuleList([])
        for i, idx in enumerate(range(len(self.depth) - 2, -1, -1)):
            d, next_d = self.depth[idx], self.depth[idx + 1]
            structure = [d for _ in range(i + 2)]
            structure_skip = [next_d for _ in range(i)]
            structure[0] += ex
            if len(structure_skip) > 0:
                structure_skip[0] += ex
            self.link_ops.append(DenseBlock(self.spatial_dimensions, structure, 3, self.adn_fn,
                                            structure_skip, True))

    def _aux_head(self, s, ex, nc):
        op = self._conv
        return torch.nn.Sequential(
            op(s, s - ex, 3, padding="same"), self.adn_fn(s - ex),
            op(s - ex, s - ex, 1, padding="same"), self.adn_fn(s - ex), op(s - ex, nc, 1))

    def init_final_layer(self):
        ex = self.skip_conditioning if self.skip_conditioning is not None else 0
        if self.n_classes > 2:
            self.final_act = torch.nn.Softmax(dim=1)
            nc = self.n_classes
        else:
            self.final_act = torch.nn.Sigmoid()
            nc = 1
        o = self.depth[0]
        self.final_layer = self._aux_head(o, 0, nc)
        S = [o + ex for _ in self.depth[:-1]]
        S[-1] = S[-1] - ex
        self.final_layer_aux = torch.nn.ModuleList([self._aux_head(s, ex, nc) for s in S])

    def _act(self, X):
        if isinstance(self.final_act, torch.nn.Sigmoid):
            if X.dim() == 4:   # 2-D network: depth-1 volume
                return HF.norm_drop_act(X.unsqueeze(2), act="sigmoid").squeeze(2)
            return HF.norm_drop_act(X, act="sigmoid")
        return HF.channel_softmax(X)   # torch.nn.Softmax(dim=1)

    def forward(self, X: torch.Tensor, return_aux=True, X_skip_layer: torch.Tensor = None,
                X_feature_conditioning: torch.Tensor = None, return_features=False,
                return_logits=False):
        if not X.is_cuda:
            raise AdellHipError("adell_mri_amd.UNetPlusPlus runs on MI355X only (no CPU fallback)")
        if X_feature_conditioning is not None:
            raise NotImplementedError("feature conditioning is outside the HIP path built so far")
        if X_skip_layer is not None and len(X_skip_layer.shape) < len(X.shape):
            X_skip_layer = X_skip_layer.unsqueeze(1)

        encoding_out, curr = [], X
        for level, downsample in self.encoding_operations:
            curr = level(curr)
            encoding_out.append(curr)
            curr = downsample(curr)
        bottleneck = curr
        # dense links: level i's DenseBlock takes the skip tensor and all but the last output
        # of the previous (coarser) level's block, upsampled inside the block
        dense = None
        for i, op in enumerate(self.decoding_operations):
            link_in = encoding_out[-i - 2]
            if X_skip_layer is not None:
                link_in = _cat_channels(link_in, _nearest(X_skip_layer, link_in.shape[2:]))
            dense = self.link_ops[i](link_in, None if dense is None else dense[:-1])
            encoded = dense[-1]
            curr = self.upscale_ops[i](curr)
            sh, sh2 = list(curr.shape)[2:], list(encoded.shape)[2:]
            if np.prod(sh) < np.prod(sh2):      # the pooling encoder leaves odd sizes behind
                encoded = crop_to_size(encoded, sh)
            if np.prod(sh) > np.prod(sh2):
                curr = crop_to_size(curr, sh2)
            curr = op(curr, X_cat=encoded)

        final_features = curr
        curr = self.final_layer(curr)
        if return_logits is False:
            curr = self._act(curr)
        if return_features is True:
            return curr, final_features, bottleneck
        curr_aux = None
        if return_aux is True:   # one auxiliary head per intermediate output of the last block
            curr_aux = []
            for head, x in zip(self.final_layer_aux, dense[1:-1]):
                if X_skip_layer is not None:
                    x = _cat_channels(x, X_skip_layer)
                curr_aux.append(self._act(head(x)))
        bn_out = None
        if self.bottleneck_classification is True:
            bn_out = self.bottleneck_classifier(HF.channel_max(bottleneck))
        return curr, bn_out, curr_aux
