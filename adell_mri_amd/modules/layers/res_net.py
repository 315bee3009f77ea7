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
from .conv import Conv3d, MaxPool3d
from .linear_blocks import LayerNorm, Linear
from .res_blocks import ResidualBlock3d


def resnet_to_encoding_ops(res_net: List[torch.nn.Module]) -> torch.nn.ModuleList:
    """U-Net ``encoding_operations`` ([[op, pool], ...] per backbone) from ResNet objects
    (res_net.py:27-48; the same repackaging is inlined at
    entrypoints/segmentation/train.py:711-733)."""
    backbone = [x.backbone for x in res_net]
    res_ops = [[x.input_layer, *x.operations] for x in backbone]
    res_pool_ops = [[x.first_pooling, *x.pooling_operations] for x in backbone]
    encoding_operations = [torch.nn.ModuleList([]) for _ in res_ops]
    for i in range(len(res_ops)):
        for a, b in zip(res_ops[i], res_pool_ops[i]):
            encoding_operations[i].append(torch.nn.ModuleList([a, b]))
    return torch.nn.ModuleList(encoding_operations)


class ResNetBackbone(torch.nn.Module):
    def __init__(self, spatial_dim: int, in_channels: int,
                 structure: List[Tuple[int, int, int, int]],
                 maxpool_structure: List[Union[Tuple[int, int], Tuple[int, int, int]]] = None,
                 padding=None, adn_fn: torch.nn.Module = torch.nn.Identity,
                 res_type: str = "resnet", batch_ensemble: int = 0,
                 skip_last_activation: bool = False):
        super().__init__()
        self.spatial_dim = spatial_dim
        self.in_channels = in_channels
        self.structure = structure
        self.maxpool_structure = maxpool_structure
        if self.maxpool_structure is None:
            self.maxpool_structure = [2 for _ in self.structure]
        self.adn_fn = adn_fn
        self.res_type = res_type
        self.batch_ensemble = batch_ensemble
        self.skip_last_activation = skip_last_activation
        if spatial_dim != 3 or res_type != "resnet" or batch_ensemble > 0:
            raise NotImplementedError("HIP ResNetBackbone covers spatial_dim=3, res_type='resnet', "
                                      "batch_ensemble=0")
        self.get_ops()
        self.init_layers()
        self.output_features = self.structure[-1][0]

    def get_ops(self):
        self.res_op = ResidualBlock3d
        self.conv_op = Conv3d
        self.max_pool_op = MaxPool3d

    def init_layers(self):
        f = self.structure[0][0]
        self.input_layer = torch.nn.Sequential(
            self.conv_op(self.in_channels, f, 7, padding="same"), self.adn_fn(f),
            self.conv_op(f, f, 3, padding="same"), self.adn_fn(f))
        self.first_pooling = self.max_pool_op(2, 2)
        self.operations = torch.nn.ModuleList([])
        self.be_operations = torch.nn.ModuleList([])
        self.pooling_operations = torch.nn.ModuleList([])
        prev_inp = f
        for s, mp in zip(self.structure, self.maxpool_structure):
            inp, inter, k, N = s
            op = [self.res_op(prev_inp, k, inter, inp, self.adn_fn)]
            for _ in range(1, N - 1):
                op.append(self.res_op(inp, k, inter, inp, self.adn_fn))
            op.append(self.res_op(inp, k, inter, inp, self.adn_fn))
            prev_inp = inp
            self.operations.append(torch.nn.Sequential(*op))
            self.be_operations.append(None)
            self.pooling_operations.append(self.max_pool_op(mp, mp))

    def forward_with_intermediate(self, X, after_pool: bool = False, batch_idx: int = None):
        X = self.first_pooling(self.input_layer(X))
        output_list = []
        for op, pool_op in zip(self.operations, self.pooling_operations):
            X = op(X)
            pooled_X = pool_op(X)
            output_list.append(pooled_X if after_pool is True else X)
            X = pooled_X
        return X, output_list

    def forward_intermediate(self, X, after_pool: bool = False, batch_idx: int = None):
        output_list = []
        X = self.input_layer(X)
        if after_pool is False:
            output_list.append(X)
        X = self.first_pooling(X)
        if after_pool is True:
            output_list.append(X)
        for op, pooling_op in zip(self.operations, self.pooling_operations):
            X = op(X)
            pooled_X = pooling_op(X)
            output_list.append(pooled_X if after_pool is True else X)
            X = pooled_X
        return output_list

    def forward_regular(self, X, batch_idx: int = None):
        X, _ = self.forward_with_intermediate(X, after_pool=False, batch_idx=batch_idx)
        return X

    def forward(self, X, return_intermediate: bool = False, after_pool: bool = False,
                batch_idx: int = None):
        if return_intermediate is True:
            return self.forward_with_intermediate(X, after_pool=after_pool)
        return self.forward_regular(X, batch_idx=batch_idx)


class ProjectionHead(torch.nn.Module):
    """Global max over the volume, then Linear (+ADN) layers (res_net.py:278-324)."""

    def __init__(self, in_channels: int, structure: List[int],
                 adn_fn: torch.nn.Module = torch.nn.Identity):
        super().__init__()
        self.in_channels = in_channels
        self.structure = structure
        self.adn_fn = adn_fn
        self.init_head()

    def init_head(self):
        prev_d = self.in_channels
        ops = OrderedDict()
        i = -1
        for i, fd in enumerate(self.structure[:-1]):
            ops["linear_{}".format(i)] = torch.nn.Sequential(Linear(prev_d, fd), self.adn_fn(fd))
            prev_d = fd
        ops["linear_{}".format(i + 1)] = Linear(prev_d, self.structure[-1])
        self.op = torch.nn.Sequential(ops)

    def forward(self, X):
        if len(X.shape) == 5:
            X = HF.max_pool3d(X, tuple(X.shape[2:]), tuple(X.shape[2:]), 0).flatten(1)
        elif len(X.shape) > 2:
            raise NotImplementedError("HIP ProjectionHead pools 5-D volumes only")
        return self.op(X)


class ResNet(torch.nn.Module):
    def __init__(self, backbone_args: dict, projection_head_args: dict = None,
                 prediction_head_args: dict = None):
        super().__init__()
        self.backbone_args = backbone_args
        self.projection_head_args = projection_head_args
        self.prediction_head_args = prediction_head_args
        self.backbone = ResNetBackbone(**self.backbone_args)
        self.init_projection_head()
        self.init_prediction_head()

    def init_projection_head(self):
        if self.projection_head_args is not None:
            args = dict(self.projection_head_args)
            d = args["structure"][-1]
            if "last_layer_norm" in args:
                norm_fn = args.pop("last_layer_norm")
            elif hasattr(args["adn_fn"](d), "norm_fn"):
                norm_fn = args["adn_fn"](d).norm_fn
            else:
                norm_fn = LayerNorm
            self.projection_head = torch.nn.Sequential(ProjectionHead(**args), _NormLeaf(norm_fn, d))

    def init_prediction_head(self):
        if self.prediction_head_args is not None:
            self.prediction_head = ProjectionHead(**self.prediction_head_args)

    def forward_representation(self, X, *args, **kwargs):
        return self.backbone(X, *args, **kwargs)

    def forward_representation_with_intermediate(self, X):
        return self.backbone.forward_with_intermediate(X)

    def forward_intermediate(self, X):
        return self.backbone.forward_intermediate(X)

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
