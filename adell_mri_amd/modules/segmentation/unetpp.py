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
from .unet import UNet


class UNetPlusPlus(UNet):
    def __init__(
        self,
        spatial_dimensions: int = 2,
        encoding_operations: List[torch.nn.ModuleList] = None,
        conv_type: str = "regular",
        link_type: str = "identity",
        upscale_type: str = "upsample",
        interpolation: str = "bilinear",
        norm_type: str = "batch",
        dropout_type: str = "dropout",
        padding: int = 0,
        dropout_param: float = 0.1,
        activation_fn: torch.nn.Module = torch.nn.PReLU,
        in_channels: int = 1,
        n_classes: int = 2,
        depth: list = [16, 32, 64],
        kernel_sizes: list = [3, 3, 3],
        strides: list = [2, 2, 2],
        bottleneck_classification: bool = False,
        skip_conditioning: int = None,
        feature_conditioning: int = None,
        feature_conditioning_params: Dict[str, torch.Tensor] = None,
    ) -> torch.nn.Module:
        super().__init__(
            spatial_dimensions=spatial_dimensions, encoding_operations=encoding_operations,
            conv_type=conv_type, upscale_type=upscale_type, interpolation=interpolation,
            norm_type=norm_type, dropout_type=dropout_type, padding=padding,
            dropout_param=dropout_param, activation_fn=activation_fn, in_channels=in_channels,
            n_classes=n_classes, depth=depth, kernel_sizes=kernel_sizes, strides=strides,
            bottleneck_classification=bottleneck_classification,
            skip_conditioning=skip_conditioning, feature_conditioning=feature_conditioning,
            feature_conditioning_params=feature_conditioning_params)
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

        encoding_out = []
        curr = X
        for op, op_ds in self.encoding_operations:
            curr = op(curr)
            encoding_out.append(curr)
            curr = op_ds(curr)
        bottleneck = curr
        link_outputs = []
        for i in range(len(self.decoding_operations)):
            op = self.decoding_operations[i]
            lo = link_outputs[-1][:-1] if len(link_outputs) > 0 else None
            link_in = encoding_out[-i - 2]
            if X_skip_layer is not None:
                xfl = HF.interpolate_nearest(X_skip_layer, link_in.shape[2:])
                link_in = HF.cat_channels([link_in, xfl])
            encoded = self.link_ops[i](link_in, lo)
            link_outputs.append(encoded)
            encoded = encoded[-1]
            curr = self.upscale_ops[i](curr)
            sh, sh2 = list(curr.shape)[2:], list(encoded.shape)[2:]
            if np.prod(sh) < np.prod(sh2):
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

        if return_aux is True:
            curr_aux = []
            for op, x in zip(self.final_layer_aux, link_outputs[-1][1:-1]):
                if X_skip_layer is not None:
                    x = HF.cat_channels([x, X_skip_layer])
                curr_aux.append(self._act(op(x)))
        else:
            curr_aux = None
        if self.bottleneck_classification is True:
            pooled = HF.channel_max(bottleneck)
            bn_out = self.bottleneck_classifier(pooled)
        else:
            bn_out = None
        return curr, bn_out, curr_aux
