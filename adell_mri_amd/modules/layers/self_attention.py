"""Squeeze-and-excite gates (mirror of adell_mri/modules/layers/self_attention.py:21-150): same
sub-module trees and `state_dict` keys (`op.0.weight`, `spatial.op.0.weight`,
`channel.op.2.bias`, ...); the arithmetic runs on the MI355X kernels. The concurrent block is one
fused pass, x * (spatial gate + channel gate), instead of two scaled copies and an add."""
import torch

from ... import functional as HF
from .conv import Conv2d, Conv3d


def _as5d(X):
    return X.unsqueeze(2) if X.dim() == 4 else X


def _like(Y, X):
    return Y.squeeze(2) if X.dim() == 4 else Y


def _spatial_gate(op, X5):
    """sigmoid(pointwise conv C -> 1) of a [N, C, D, H, W] tensor: [N, 1, D, H, W]."""
    conv = op[0]
    w = conv.weight if conv.weight.dim() == 5 else conv.weight.unsqueeze(2)
    return HF.norm_drop_act(HF.conv3d(X5, w, conv.bias, 1, 0, want_stats=False), act="sigmoid")


def _channel_gate(op, X5):
    """sigmoid(Linear(relu(Linear(mean over voxels)))): [N, C]."""
    h = HF.elementwise(HF.linear(HF.channel_mean(X5), op[0].weight, op[0].bias), act="relu")
    return HF.elementwise(HF.linear(h, op[2].weight, op[2].bias), act="sigmoid")


class _SpatialSqueezeAndExcite(torch.nn.Module):
    _conv = None

    def __init__(self, input_channels: int):
        super().__init__()
        self.input_channels = input_channels
        self.init_layers()

    def init_layers(self):
        self.op = torch.nn.Sequential(type(self)._conv(self.input_channels, 1, kernel_size=1),
                                      torch.nn.Sigmoid())

    def forward(self, X: torch.Tensor) -> torch.Tensor:
        X5 = _as5d(X)
        zeros = X5.new_zeros((X5.shape[0], X5.shape[1]))
        return _like(HF.cse_apply(X5, _spatial_gate(self.op, X5), zeros), X)


class SpatialSqueezeAndExcite2d(_SpatialSqueezeAndExcite):
    """self_attention.py:21-52."""
    _conv = Conv2d


class SpatialSqueezeAndExcite3d(_SpatialSqueezeAndExcite):
    """self_attention.py:55-86."""
    _conv = Conv3d


class ChannelSqueezeAndExcite(torch.nn.Module):
    """self_attention.py:89-124."""

    def __init__(self, input_channels: int):
        super().__init__()
        self.input_channels = input_channels
        self.init_layers()

    def init_layers(self):
        n_chan = self.input_channels
        self.op = torch.nn.Sequential(torch.nn.Linear(n_chan, n_chan), torch.nn.ReLU(),
                                      torch.nn.Linear(n_chan, n_chan), torch.nn.Sigmoid())

    def forward(self, X: torch.Tensor) -> torch.Tensor:
        X5 = _as5d(X)
        return _like(HF.scale_per_item_channel(X5, _channel_gate(self.op, X5)), X)


class _ConcurrentSqueezeAndExcite(torch.nn.Module):
    _spatial = None

    def __init__(self, input_channels: int):
        super().__init__()
        self.input_channels = input_channels
        self.init_layers()

    def init_layers(self):
        self.spatial = type(self)._spatial(self.input_channels)
        self.channel = ChannelSqueezeAndExcite(self.input_channels)

    def forward(self, X, inv=None, acc=None):
        """spatial(X) + channel(X) = X * (s + c); ``inv`` [N] and ``acc`` fold the per-item
        division and the running sum over branches of BrUNet.forward into the same pass."""
        X5 = _as5d(X)
        s = _spatial_gate(self.spatial.op, X5)
        c = _channel_gate(self.channel.op, X5)
        return _like(HF.cse_apply(X5, s, c, inv, None if acc is None else _as5d(acc)), X)


class ConcurrentSqueezeAndExcite2d(_ConcurrentSqueezeAndExcite):
    """self_attention.py:108-125."""
    _spatial = SpatialSqueezeAndExcite2d


class ConcurrentSqueezeAndExcite3d(_ConcurrentSqueezeAndExcite):
    """self_attention.py:127-149."""
    _spatial = SpatialSqueezeAndExcite3d
