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
