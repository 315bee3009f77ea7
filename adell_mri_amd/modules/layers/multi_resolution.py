"""Multi-resolution blocks (mirror of adell_mri/modules/layers/multi_resolution.py): the atrous
spatial pyramid of the U-Net's ``conv_type="asp"`` encoder ops (unet.py:399-413)."""
from typing import List

import torch

from ... import functional as HF
from .conv import Conv3d
from .standard_blocks import DepthWiseSeparableConvolution3d
from .utils import split_int_into_n


class AtrousSpatialPyramidPooling3d(torch.nn.Module):
    """Per dilation rate: ``Conv3d(in, c_i, 3, dilation=rate, padding="same")`` -> ADN ->
    depthwise-separable 3x3x3 conv -> ADN; the outputs concatenated along the channels
    (multi_resolution.py:359-416; ``c_i`` = ``split_int_into_n(out_channels, len(rates))``).
    The dilated conv runs the ordinary conv kernels on the rate^3 interleaved sub-lattices of its
    input (functional.conv3d_dilated); same module tree and parameter names as the reference."""

    def __init__(self, in_channels: int, out_channels: int, rates: List[int],
                 adn_fn=torch.nn.Identity):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.rates = rates
        self.adn_fn = adn_fn
        self.n_channels = split_int_into_n(out_channels, len(rates))
        self.layers = torch.nn.ModuleList([
            torch.nn.Sequential(
                Conv3d(in_channels, c, kernel_size=3, dilation=rate, padding="same"),
                adn_fn(c),
                DepthWiseSeparableConvolution3d(c, c, kernel_size=3, padding="same"),
                adn_fn(c))
            for rate, c in zip(rates, self.n_channels)])

    def forward(self, X: torch.Tensor) -> torch.Tensor:
        return HF.cat_channels([layer(X) for layer in self.layers])


class AtrousSpatialPyramidPooling2d(torch.nn.Module):
    """The 2-D pyramid cannot be constructed in the reference itself
    (``DepthWiseSeparableConvolution2d.init_layers`` reads ``self.paddign``, standard_blocks.py:78:
    AttributeError), so there is no behaviour to mirror."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        raise NotImplementedError(
            "AtrousSpatialPyramidPooling2d: the reference's own constructor raises (standard_blocks.py:78 "
            "reads self.paddign); conv_type='asp' exists for spatial_dimensions=3 only")
