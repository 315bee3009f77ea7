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



class SelfAttentionBlock(torch.nn.Module):
    """Self-attention over the patches of an image / volume (self_attention.py:152-239; the U-Net's
    ``link_type="attention"`` skip links, unet.py:473-481): patches of ``patch_size`` become tokens
    of (x y z c) features -- "n c (h x) (w y) (d z) -> n (h w d) (x y z c)" -- one
    MultiHeadSelfAttention (QK-norm, 4 heads) maps them back to the same width, and the inverse
    rearrangement restores the tensor. Same sub-module tree / ``state_dict`` keys
    (``attention_op.qkv.weight``, ...). The rearrangements are index permutations (a view plus one
    copy); the arithmetic is the MHSA kernels."""

    def __init__(self, ndim: int, input_dim: int, attention_dim: int, patch_size=(16, 16, 8)):
        super().__init__()
        import numpy as np

        from .linear_blocks import MultiHeadSelfAttention

        self.ndim = ndim
        self.input_dim = input_dim
        self.attention_dim = attention_dim
        self.patch_size = patch_size
        self.input_dim_att = int(np.prod(patch_size[:ndim]) * input_dim)
        self.attention_op = MultiHeadSelfAttention(
            input_dim=self.input_dim_att, attention_dim=attention_dim, hidden_dim=attention_dim,
            output_dim=self.input_dim_att)

    def _grid(self, shape):
        ps = list(self.patch_size[:self.ndim])
        for n, p in zip(shape[2:], ps):
            if n % p:
                raise ValueError(f"SelfAttentionBlock: size {tuple(shape[2:])} is not a multiple "
                                 f"of the patch size {tuple(ps)}")
        return [n // p for n, p in zip(shape[2:], ps)], ps

    def embed(self, X: torch.Tensor) -> torch.Tensor:
        n, c = X.shape[:2]
        hs, ps = self._grid(X.shape)
        shape = [n, c]
        for h, p in zip(hs, ps):
            shape += [h, p]
        nd = self.ndim
        grid_axes = [2 + 2 * i for i in range(nd)]
        patch_axes = [3 + 2 * i for i in range(nd)]
        tokens = X.reshape(shape).permute(0, *grid_axes, *patch_axes, 1)
        t = 1
        for h in hs:
            t *= h
        return tokens.reshape(n, t, self.input_dim_att)

    def unembed(self, X: torch.Tensor, sh) -> torch.Tensor:
        n, c = sh[:2]
        hs, ps = self._grid(sh)
        nd = self.ndim
        X = X.reshape([n, *hs, *ps, c])
        # axes: 0 n | 1..nd grid | nd+1..2nd patch | last c  ->  n c (h x) (w y) (d z)
        perm = [0, 1 + 2 * nd]
        for i in range(nd):
            perm += [1 + i, 1 + nd + i]
        return X.permute(perm).reshape(sh)

    def forward(self, X: torch.Tensor) -> torch.Tensor:
        return self.unembed(self.attention_op(self.embed(X)), tuple(X.shape))
