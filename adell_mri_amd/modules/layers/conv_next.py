"""ConvNeXt on the HIP kernels: drop-in for ``adell_mri/modules/layers/conv_next.py:86-235``
(backbone) and ``:388-452`` (backbone + heads), 2-D (depth-1 volumes) and 3-D. The boundary is the reference's
constructor signatures, attribute names and ``state_dict`` keys (``input_layer.{0,1}``,
``operations.<stage>.<block>``, ``projection_head.{0,1}``, ``prediction_head``); everything behind
it is this package's layers -- depthwise stencil + LDS-tiled pointwise GEMMs inside
``ConvNeXtBlock3d``, channels-first LayerNorm kernel, max-pool kernel."""
from typing import List, Tuple, Union

import torch

from .conv import Conv2d, Conv3d, MaxPool2d, MaxPool3d
from .regularization import LayerNorm
from .res_blocks import ConvNeXtBlock2d, ConvNeXtBlock3d
from .res_net import ProjectionHead, _NormLeaf


def _stage(block, width_in: int, width: int, inner: int, kernel: int, n_blocks: int):
    """One resolution level: ``n_blocks`` ConvNeXt blocks (never fewer than two, as the reference
    builds them), the first of which changes the channel count."""
    widths = [width_in] + [width] * max(n_blocks - 1, 1)
    return torch.nn.Sequential(*[block(w, kernel, inner, width) for w in widths])


class ConvNeXtBackbone(torch.nn.Module):
    def __init__(self, spatial_dim: int, in_channels: int,
                 structure: List[Tuple[int, int, int, int]],
                 maxpool_structure: List[Union[Tuple[int, int], Tuple[int, int, int]]] = None,
                 first_layer_stride=4, padding=None, adn_fn: torch.nn.Module = torch.nn.Identity,
                 batch_ensemble: int = 0):
        super().__init__()
        if spatial_dim not in (2, 3) or batch_ensemble > 0:
            raise NotImplementedError("HIP ConvNeXtBackbone covers spatial_dim 2 / 3, "
                                      "batch_ensemble=0")
        if maxpool_structure is None:
            maxpool_structure = [2] * len(structure)
        self.spatial_dim, self.in_channels = spatial_dim, in_channels
        self.structure, self.maxpool_structure = structure, maxpool_structure
        self.first_layer_stride, self.adn_fn, self.batch_ensemble = (first_layer_stride, adn_fn,
                                                                     batch_ensemble)
        self.res_op, self.conv_op, self.max_pool_op = (
            (ConvNeXtBlock2d, Conv2d, MaxPool2d) if spatial_dim == 2
            else (ConvNeXtBlock3d, Conv3d, MaxPool3d))
        stem_width = structure[0][0]
        # patchify stem: k = 4 conv at the stem stride, then LayerNorm over channels per voxel
        self.input_layer = torch.nn.Sequential(
            self.conv_op(in_channels, stem_width, 4, stride=first_layer_stride),
            LayerNorm(stem_width, data_format="channels_first"))
        stages, pools, width_in = [], [], stem_width
        for (width, inner, kernel, n_blocks), pool in zip(structure, maxpool_structure):
            stages.append(_stage(self.res_op, width_in, width, inner, kernel, n_blocks))
            pools.append(self.max_pool_op(pool, pool))
            width_in = width
        self.operations = torch.nn.ModuleList(stages)
        self.be_operations = torch.nn.ModuleList([None] * len(stages))  # batch-ensemble slots: unused
        self.pooling_operations = torch.nn.ModuleList(pools)
        self.output_features = structure[-1][0]
        for m in self.modules():   # trunc-normal(0.02) weights, zero biases on convs and linears
            if isinstance(m, (torch.nn.Conv2d, torch.nn.Conv3d, torch.nn.Linear)):
                torch.nn.init.trunc_normal_(m.weight, std=0.02)
                torch.nn.init.constant_(m.bias, 0)

    def _walk(self, X):
        """Yields (stage output, pooled stage output) level by level."""
        X = self.input_layer(X)
        for stage, pool in zip(self.operations, self.pooling_operations):
            features = stage(X)
            X = pool(features)
            yield features, X

    def forward_with_intermediate(self, X, after_pool=False):
        kept, last = [], None
        for features, pooled in self._walk(X):
            kept.append(pooled if after_pool else features)
            last = pooled
        return last, kept

    def forward_regular(self, X, batch_idx=None):
        for _, X in self._walk(X):
            pass
        return X

    def forward(self, X, return_intermediate: bool = False, after_pool: bool = False,
                batch_idx: bool = None):
        if return_intermediate is True:
            return self.forward_with_intermediate(X, after_pool=after_pool)
        return self.forward_regular(X, batch_idx=batch_idx)


class ConvNeXt(torch.nn.Module):
    """Backbone + projection head (+ normalisation leaf) + optional prediction head."""

    def __init__(self, backbone_args: dict, projection_head_args: dict,
                 prediction_head_args: dict = None):
        super().__init__()
        self.backbone_args = backbone_args
        self.projection_head_args = projection_head_args
        self.prediction_head_args = prediction_head_args
        self.backbone = ConvNeXtBackbone(**backbone_args)
        self.projection_head = self._projection(projection_head_args)
        if prediction_head_args is not None:
            self.prediction_head = ProjectionHead(**prediction_head_args)

    @staticmethod
    def _projection(head_args):
        if head_args is None:
            return torch.nn.Identity()
        head_args = dict(head_args)
        width = head_args["structure"][-1]
        try:      # the normalisation the head's own ADN uses; LayerNorm when it has none
            norm_fn = head_args["adn_fn"](width).norm_fn
        except Exception:  # noqa: BLE001
            norm_fn = torch.nn.LayerNorm
        return torch.nn.Sequential(ProjectionHead(**head_args), _NormLeaf(norm_fn, width))

    def forward_representation(self, X, *args, **kwargs):
        return self.backbone(X, *args, **kwargs)

    def forward_representation_with_intermediate(self, X):
        return self.backbone.forward_with_intermediate(X)

    def forward(self, X, ret="projection"):
        """``ret``: "representation" (backbone output), "projection" or "prediction"."""
        out = self.backbone(X)
        if ret != "representation":
            out = self.projection_head(out)
            if ret == "prediction":
                out = self.prediction_head(out)
            elif ret != "projection":
                return None
        return out
