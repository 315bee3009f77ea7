"""ConvNeXt backbone and network (mirror of adell_mri/modules/layers/conv_next.py:86-235,
388-452) in 3-D: patchify stem (Conv3d k=4, stride first_layer_stride) + channels-first
LayerNorm, stages of ConvNeXtBlock3d followed by MaxPool3d, projection / prediction
heads. Same constructors, module tree and state_dict keys."""
from typing import List, Tuple, Union

import torch

from .conv import Conv3d, MaxPool3d
from .linear_blocks import LayerNorm as RowLayerNorm
from .regularization import LayerNorm
from .res_blocks import ConvNeXtBlock3d
from .res_net import ProjectionHead, _NormLeaf


class ConvNeXtBackbone(torch.nn.Module):
    def __init__(self, spatial_dim: int, in_channels: int,
                 structure: List[Tuple[int, int, int, int]],
                 maxpool_structure: List[Union[Tuple[int, int], Tuple[int, int, int]]] = None,
                 first_layer_stride=4, padding=None, adn_fn: torch.nn.Module = torch.nn.Identity,
                 batch_ensemble: int = 0):
        super().__init__()
        self.spatial_dim = spatial_dim
        self.in_channels = in_channels
        self.structure = structure
        self.maxpool_structure = maxpool_structure
        self.first_layer_stride = first_layer_stride
        if self.maxpool_structure is None:
            self.maxpool_structure = [2 for _ in self.structure]
        self.adn_fn = adn_fn
        self.batch_ensemble = batch_ensemble
        if spatial_dim != 3 or batch_ensemble > 0:
            raise NotImplementedError("HIP ConvNeXtBackbone covers spatial_dim=3, batch_ensemble=0")
        self.get_ops()
        self.init_layers()
        self.output_features = self.structure[-1][0]

    def get_ops(self):
        self.res_op = ConvNeXtBlock3d
        self.conv_op = Conv3d
        self.max_pool_op = MaxPool3d

    def init_input_layer(self):
        f = self.structure[0][0]
        return torch.nn.Sequential(
            self.conv_op(self.in_channels, f, 4, stride=self.first_layer_stride),
            LayerNorm(f, data_format="channels_first"))

    def init_layers(self):
        f = self.structure[0][0]
        self.input_layer = self.init_input_layer()
        self.operations = torch.nn.ModuleList([])
        self.be_operations = torch.nn.ModuleList([])
        self.pooling_operations = torch.nn.ModuleList([])
        prev_inp = f
        for s, mp in zip(self.structure, self.maxpool_structure):
            inp, inter, k, N = s
            op = [self.res_op(prev_inp, k, inter, inp)]
            for _ in range(1, N - 1):
                op.append(self.res_op(inp, k, inter, inp))
            op.append(self.res_op(inp, k, inter, inp))
            prev_inp = inp
            self.operations.append(torch.nn.Sequential(*op))
            self.be_operations.append(None)
            self.pooling_operations.append(self.max_pool_op(mp, mp))
        self.apply(self._init_weights)

    def _init_weights(self, m):
        if isinstance(m, (torch.nn.Conv3d, torch.nn.Linear)):
            torch.nn.init.trunc_normal_(m.weight, std=0.02)
            torch.nn.init.constant_(m.bias, 0)

    def forward_with_intermediate(self, X, after_pool=False):
        X = self.input_layer(X)
        output_list = []
        for op, pool_op in zip(self.operations, self.pooling_operations):
            if after_pool is False:
                X = op(X)
                output_list.append(X)
                X = pool_op(X)
            else:
                X = pool_op(op(X))
                output_list.append(X)
        return X, output_list

    def forward_regular(self, X, batch_idx=None):
        X = self.input_layer(X)
        for op, pool_op in zip(self.operations, self.pooling_operations):
            X = pool_op(op(X))
        return X

    def forward(self, X, return_intermediate: bool = False, after_pool: bool = False,
                batch_idx: bool = None):
        if return_intermediate is True:
            return self.forward_with_intermediate(X, after_pool=after_pool)
        return self.forward_regular(X, batch_idx=batch_idx)


class ConvNeXt(torch.nn.Module):
    def __init__(self, backbone_args: dict, projection_head_args: dict,
                 prediction_head_args: dict = None):
        super().__init__()
        self.backbone_args = backbone_args
        self.projection_head_args = projection_head_args
        self.prediction_head_args = prediction_head_args
        self.backbone = ConvNeXtBackbone(**self.backbone_args)
        self.init_projection_head()
        self.init_prediction_head()

    def init_projection_head(self):
        if self.projection_head_args is not None:
            args = dict(self.projection_head_args)
            try:
                d = args["structure"][-1]
                norm_fn = args["adn_fn"](d).norm_fn
            except Exception:  # noqa: BLE001 -- same fallback as the reference
                norm_fn = torch.nn.LayerNorm
            self.projection_head = torch.nn.Sequential(ProjectionHead(**args), _NormLeaf(norm_fn, d))
        else:
            self.projection_head = torch.nn.Identity()

    def init_prediction_head(self):
        if self.prediction_head_args is not None:
            self.prediction_head = ProjectionHead(**self.prediction_head_args)

    def forward_representation(self, X, *args, **kwargs):
        return self.backbone(X, *args, **kwargs)

    def forward_representation_with_intermediate(self, X):
        return self.backbone.forward_with_intermediate(X)

    def forward(self, X, ret="projection"):
        X = self.backbone(X)
        if ret == "representation":
            return X
        X = self.projection_head(X)
        if ret == "projection":
            return X
        X = self.prediction_head(X)
        if ret == "prediction":
            return X
