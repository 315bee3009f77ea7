"""Dense block of the U-Net++ skip links (mirror of
adell_mri/modules/layers/standard_blocks.py:284-376): same constructor, ``ops``
tree and forward semantics. The growing concatenation
``cat([out, *outputs[:-1], *skips])`` is assembled by the HIP channel-copy kernel
for all but its last member, which the conv reads as its second (virtual) source;
the resolution change of the skip inputs is the HIP nearest-neighbour resampler.
"""
from typing import List

import torch

from ... import functional as HF
from .conv import Conv2d, Conv3d


class DenseBlock(torch.nn.Module):
    def __init__(self, spatial_dim: int, structure: List[int], kernel_size: int,
                 adn_fn: torch.nn.Module = torch.nn.PReLU, structure_skip: List[int] = None,
                 return_all: bool = False):
        super().__init__()
        self.spatial_dim = spatial_dim
        self.structure = structure
        self.kernel_size = kernel_size
        self.adn_fn = adn_fn
        self.structure_skip = structure_skip
        self.return_all = return_all
        if self.structure_skip is None or len(self.structure_skip) == 0:
            self.structure_skip = [0 for _ in range(len(self.structure) - 1)]
        self.init_layers()

    def init_layers(self):
        self.conv_op = Conv2d if self.spatial_dim == 2 else Conv3d
        self.ops = torch.nn.ModuleList([])
        self.upscale_ops = torch.nn.ModuleList([])
        k = self.kernel_size
        self.ops.append(torch.nn.Sequential(
            self.conv_op(self.structure[0], self.structure[1], k, padding="same"),
            self.adn_fn(self.structure[1])))
        for i in range(1, len(self.structure) - 1):
            prev_d = sum(self.structure[:(i + 1)]) + self.structure_skip[i - 1]
            d = self.structure[i + 1]
            self.ops.append(torch.nn.Sequential(self.conv_op(prev_d, d, k, padding="same"),
                                                self.adn_fn(d)))

    def forward(self, X: torch.Tensor, X_skip=None):
        outputs = [X]
        out = X
        for i in range(len(self.ops)):
            xs = []
            if X_skip is not None and i > 0:
                xs = [HF.interpolate_nearest(X_skip[i - 1], out.shape[2:])]
            members = [out, *outputs[:-1], *xs]
            conv, adn = self.ops[i][0], self.ops[i][1]
            if len(members) == 1:
                out = conv(members[0])
            else:
                out = conv(HF.cat_channels(members[:-1]), X_cat=members[-1])
            out = adn(out)
            outputs.append(out)
        return outputs if self.return_all is True else outputs[-1]


class DepthWiseSeparableConvolution3d(torch.nn.Module):
    """Depthwise ``Conv3d(c, c, k, padding, groups=c)`` -> pointwise ``Conv3d(c, out, 1)`` -> ADN
    (adell_mri/modules/layers/standard_blocks.py:93-144): the stencil kernel and the 1x1x1 conv on the
    GEMM / implicit-GEMM kernels. Same attribute names (``depthwise_op``, ``pointwise_op``,
    ``act_op``) and parameter shapes."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int = 3, padding=1,
                 adn_fn=torch.nn.Identity):
        super().__init__()
        from .res_blocks import DepthwiseConv3d

        self.input_channels = in_channels
        self.output_channels = out_channels
        self.kernel_size = kernel_size
        self.padding = padding
        self.adn_fn = adn_fn
        if padding != "same" and tuple(torch.nn.modules.utils._triple(padding)) != \
                tuple(k // 2 for k in torch.nn.modules.utils._triple(kernel_size)):
            raise NotImplementedError("HIP DepthWiseSeparableConvolution3d: padding 'same' (or k // 2)")
        self.depthwise_op = DepthwiseConv3d(in_channels, in_channels, kernel_size=kernel_size,
                                            padding="same", groups=in_channels)
        self.pointwise_op = Conv3d(in_channels, out_channels, kernel_size=1)
        self.act_op = adn_fn(out_channels)

    def forward(self, X: torch.Tensor) -> torch.Tensor:
        return self.act_op(self.pointwise_op(self.depthwise_op(X)))
