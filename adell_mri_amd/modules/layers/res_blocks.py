"""Residual blocks (mirror of adell_mri/modules/layers/res_blocks.py:108-200).

Same module tree (``op``, ``final_op``, ``adn_op``) and ``state_dict`` keys; the
skip addition ``op(X) + X`` is not a separate pass: it is the ``residual``
operand of the last convolution's epilogue, which also reduces the statistics
``adn_op``'s instance/batch norm needs.
"""
import torch

from .conv import Conv3d


class ResidualBlock3d(torch.nn.Module):
    def __init__(self, in_channels: int, kernel_size: int, inter_channels: int = None,
                 out_channels: int = None, adn_fn: torch.nn.Module = torch.nn.Identity,
                 skip_activation: bool = None):
        super().__init__()
        self.in_channels = in_channels
        self.kernel_size = kernel_size
        self.inter_channels = inter_channels
        self.out_channels = out_channels if out_channels is not None else in_channels
        self.adn_fn = adn_fn
        self.skip_activation = skip_activation
        self.init_layers()

    def init_layers(self):
        c, k, m = self.in_channels, self.kernel_size, self.inter_channels
        if m is not None:
            self.op = torch.nn.Sequential(
                Conv3d(c, m, 1), self.adn_fn(m), Conv3d(m, m, k, padding="same"),
                self.adn_fn(m), Conv3d(m, c, 1))
        else:
            self.op = torch.nn.Sequential(
                Conv3d(c, c, k, padding="same"), self.adn_fn(c), Conv3d(c, c, k, padding="same"))
        if self.in_channels != self.out_channels:
            self.final_op = Conv3d(self.in_channels, self.out_channels, 1)
        else:
            self.final_op = torch.nn.Identity()
        self.adn_op = self.adn_fn(self.out_channels)

    def forward(self, X: torch.Tensor, skip_activation: bool = None):
        h = X
        mods = list(self.op)
        for mod in mods[:-1]:
            h = mod(h)
        out = self.final_op(mods[-1](h, residual=X))
        skip = skip_activation if skip_activation is not None else self.skip_activation
        if skip is not True:
            out = self.adn_op(out)
        return out
