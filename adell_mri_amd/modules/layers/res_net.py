"""ResNet backbone, projection head and the backbone -> U-Net encoder glue (mirror
of adell_mri/modules/layers/res_net.py:27-396) for ``res_type="resnet"`` in 3-D:
same constructors, module tree / state_dict keys (``input_layer``, ``first_pooling``,
``operations``, ``pooling_operations``, ``op.linear_N``) and forward variants. Convs
(7^3 stem, k=5 / k=3 bottleneck blocks), pooling, ADN and Linear layers are the HIP
leaves.
"""
from typing import List, OrderedDict, Tuple, Union

import torch

from ... import functional as HF
from .conv import Conv2d, Conv3d, MaxPool2d, MaxPool3d
from .linear_blocks import LayerNorm, Linear
from .res_blocks import ResidualBlock2d, ResidualBlock3d


def resnet_to_encoding_ops(res_net: List[torch.nn.Module]) -> torch.nn.ModuleList:
    """U-Net ``encoding_operations`` from ResNet objects (res_net.py:27-48; inlined again at
    entrypoints/segmentation/train.py:711-733): per network, the list of [level, pooling] pairs
    -- the stem with its pooling first, then every residual stage with its own."""
    per_network = []
    for backbone in (net.backbone for net in res_net):
        levels = [backbone.input_layer, *backbone.operations]
        pools = [backbone.first_pooling, *backbone.pooling_operations]
        per_network.append(torch.nn.ModuleList(
            [torch.nn.ModuleList([level, pool]) for level, pool in zip(levels, pools)]))
    return torch.nn.ModuleList(per_network)


class ResNetBackbone(torch.nn.Module):
    """Stem (7^3 conv + ADN + 3^3 conv + ADN, max-pool 2) and residual stages of bottleneck
    blocks, each followed by a max-pool (res_net.py:51-275). 2-D or 3-D, ``res_type="resnet"``."""

    def __init__(self, spatial_dim: int, in_channels: int,
                 structure: List[Tuple[int, int, int, int]],
                 maxpool_structure: List[Union[Tuple[int, int], Tuple[int, int, int]]] = None,
                 padding=None, adn_fn: torch.nn.Module = torch.nn.Identity,
                 res_type: str = "resnet", batch_ensemble: int = 0,
                 skip_last_activation: bool = False):
        super().__init__()
        if spatial_dim not in (2, 3) or res_type != "resnet" or batch_ensemble > 0:
            raise NotImplementedError("HIP ResNetBackbone covers spatial_dim 2 / 3, "
                                      "res_type='resnet', batch_ensemble=0")
        if maxpool_structure is None:
            maxpool_structure = [2] * len(structure)
        self.spatial_dim, self.in_channels, self.structure = spatial_dim, in_channels, structure
        self.maxpool_structure, self.adn_fn, self.res_type = maxpool_structure, adn_fn, res_type
        self.batch_ensemble, self.skip_last_activation = batch_ensemble, skip_last_activation
        block, conv, pool_op = ((ResidualBlock2d, Conv2d, MaxPool2d) if spatial_dim == 2
                                else (ResidualBlock3d, Conv3d, MaxPool3d))   # 2-D: depth-1 volumes
        self.res_op, self.conv_op, self.max_pool_op = block, conv, pool_op
        stem = structure[0][0]
        self.input_layer = torch.nn.Sequential(
            conv(in_channels, stem, 7, padding="same"), adn_fn(stem),
            conv(stem, stem, 3, padding="same"), adn_fn(stem))
        self.first_pooling = pool_op(2, 2)
        stages, pools, width_in = [], [], stem
        for (width, inner, kernel, n_blocks), pool in zip(structure, maxpool_structure):
            # at least two blocks per stage, the first one changes the width
            widths = [width_in] + [width] * max(n_blocks - 1, 1)
            stages.append(torch.nn.Sequential(
                *[block(w, kernel, inner, width, adn_fn) for w in widths]))
            pools.append(pool_op(pool, pool))
            width_in = width
        self.operations = torch.nn.ModuleList(stages)
        self.be_operations = torch.nn.ModuleList([None] * len(stages))  # batch-ensemble slots: unused
        self.pooling_operations = torch.nn.ModuleList(pools)
        self.output_features = structure[-1][0]

    def _walk(self, X):
        """Yields (level output, pooled level output): the stem first, then the stages."""
        X = self.input_layer(X)
        pooled = self.first_pooling(X)
        yield X, pooled
        for stage, pool in zip(self.operations, self.pooling_operations):
            X = stage(pooled)
            pooled = pool(X)
            yield X, pooled

    def forward_intermediate(self, X, after_pool: bool = False, batch_idx: int = None):
        """Every level including the stem."""
        return [pooled if after_pool else features for features, pooled in self._walk(X)]

    def forward_with_intermediate(self, X, after_pool: bool = False, batch_idx: int = None):
        """(output, the residual stages' tensors) -- the stem is not listed."""
        levels = list(self._walk(X))
        return levels[-1][1], [pooled if after_pool else features
                               for features, pooled in levels[1:]]

    def forward_regular(self, X, batch_idx: int = None):
        for _, X in self._walk(X):
            pass
        return X

    def forward(self, X, return_intermediate: bool = False, after_pool: bool = False,
                batch_idx: int = None):
        if return_intermediate is True:
            return self.forward_with_intermediate(X, after_pool=after_pool)
        return self.forward_regular(X, batch_idx=batch_idx)


class ProjectionHead(torch.nn.Module):
    """Global max over the volume, then ``linear_i`` = Linear + ADN for every width but the last,
    which is a bare Linear (res_net.py:278-324)."""

    def __init__(self, in_channels: int, structure: List[int],
                 adn_fn: torch.nn.Module = torch.nn.Identity):
        super().__init__()
        self.in_channels, self.structure, self.adn_fn = in_channels, structure, adn_fn
        layers, width_in = OrderedDict(), in_channels
        for i, width in enumerate(structure):
            fc = Linear(width_in, width)
            last = i == len(structure) - 1
            layers[f"linear_{i}"] = fc if last else torch.nn.Sequential(fc, adn_fn(width))
            width_in = width
        self.op = torch.nn.Sequential(layers)

    def forward(self, X):
        if X.dim() == 4:   # 2-D feature map: depth-1 volume
            X = X.unsqueeze(2)
        if X.dim() == 5:
            X = HF.max_pool3d(X, tuple(X.shape[2:]), tuple(X.shape[2:]), 0).flatten(1)
        elif X.dim() > 2:
            raise NotImplementedError("HIP ProjectionHead pools 5-D volumes only")
        return self.op(X)


class ResNet(torch.nn.Module):
    """Backbone + optional projection head (closed by a normalisation) + optional prediction
    head (res_net.py:327-396)."""

    def __init__(self, backbone_args: dict, projection_head_args: dict = None,
                 prediction_head_args: dict = None):
        super().__init__()
        self.backbone_args = backbone_args
        self.projection_head_args = projection_head_args
        self.prediction_head_args = prediction_head_args
        self.backbone = ResNetBackbone(**backbone_args)
        if projection_head_args is not None:
            head_args = dict(projection_head_args)
            width = head_args["structure"][-1]
            # closing normalisation: explicit, else the one of the head's ADN, else LayerNorm
            norm_fn = head_args.pop("last_layer_norm", None)
            if norm_fn is None:
                norm_fn = getattr(head_args["adn_fn"](width), "norm_fn", LayerNorm)
            self.projection_head = torch.nn.Sequential(ProjectionHead(**head_args),
                                                       _NormLeaf(norm_fn, width))
        if prediction_head_args is not None:
            self.prediction_head = ProjectionHead(**prediction_head_args)

    def forward_representation(self, X, *args, **kwargs):
        return self.backbone(X, *args, **kwargs)

    def forward_representation_with_intermediate(self, X):
        return self.backbone.forward_with_intermediate(X)

    def forward_intermediate(self, X):
        return self.backbone.forward_intermediate(X)

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


def _NormLeaf(norm_fn, d):
    """The normalisation closing a projection head, as a HIP leaf with the state_dict
    layout of the torch class it stands for."""
    from .adn_fn import ActDropNorm

    if norm_fn in (torch.nn.LayerNorm, LayerNorm):
        return LayerNorm(d)
    if norm_fn is torch.nn.Identity:
        return torch.nn.Identity()

    class _Norm(norm_fn):  # BatchNorm1d / InstanceNorm1d...: parameters of the torch class
        def forward(self, X):
            adn = ActDropNorm.__new__(ActDropNorm)
            torch.nn.Module.__init__(adn)
            adn.training = self.training
            adn.op_list = {"normalization": self}
            adn._stages = [{"N": "normalization"}]
            return adn._run_stage(X, adn._stages[0])

    return _Norm(d)
