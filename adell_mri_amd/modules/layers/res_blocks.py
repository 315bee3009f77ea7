"""Residual blocks (mirror of adell_mri/modules/layers/res_blocks.py:108-200).

Same module tree (``op``, ``final_op``, ``adn_op``) and ``state_dict`` keys; the
skip addition ``op(X) + X`` is not a separate pass: it is the ``residual``
operand of the last convolution's epilogue, which also reduces the statistics
``adn_op``'s instance/batch norm needs.
"""
import torch

from ... import functional as HF
from ... import ops
from .adn_fn import ActDropNorm
from .conv import Conv2d, Conv3d
from .linear_blocks import LayerNorm as RowLayerNorm
from .linear_blocks import Linear


class ResidualBlock3d(torch.nn.Module):
    _conv = Conv3d   # ResidualBlock2d swaps in the 2-D convolution

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
        conv = self._conv
        if m is not None:
            self.op = torch.nn.Sequential(
                conv(c, m, 1), self.adn_fn(m), conv(m, m, k, padding="same"),
                self.adn_fn(m), conv(m, c, 1))
        else:
            self.op = torch.nn.Sequential(
                conv(c, c, k, padding="same"), self.adn_fn(c), conv(c, c, k, padding="same"))
        if self.in_channels != self.out_channels:
            self.final_op = conv(self.in_channels, self.out_channels, 1)
        else:
            self.final_op = torch.nn.Identity()
        self.adn_op = self.adn_fn(self.out_channels)

    def forward(self, X: torch.Tensor, skip_activation: bool = None, fork=None):
        """``fork``: functional.GradCarry of a U-Net skip fork -- the block's gradient with respect
        to X is left there for the other consumer of X (segmentation/unet.py); ignored (autograd adds
        the two gradients) when the block runs without its own carry."""
        mods = list(self.op)
        # the link's gradient rides the head conv's backward-data epilogue (functional.GradCarry)
        carry = (HF.GradCarry() if (self._conv is Conv3d and X.requires_grad
                                    and torch.is_grad_enabled()
                                    and not ops.FLAGS["no_grad_carry"]
                                    and not HF.grad_observed(X)) else None)
        h = mods[0](X, carry_in=carry, carry_x0=fork) if carry is not None else mods[0](X)
        for i, mod in enumerate(mods[1:-1], 1):
            # an ADN output read by the next conv only (functional.single_use; it may be written
            # as split rows: functional.expect_rows)
            only_reader = isinstance(mod, ActDropNorm) and type(mods[i + 1]) is self._conv and h.dim() == 5
            if only_reader and self._conv is Conv3d:
                HF.expect_rows(mod, mods[i + 1])
            h = mod(h)
            if only_reader:
                h = HF.single_use(h)
        out = self.final_op(mods[-1](h, residual=X, carry_out=carry) if carry is not None
                            else mods[-1](h, residual=X))
        skip = skip_activation if skip_activation is not None else self.skip_activation
        if skip is not True:
            out = self.adn_op(out)
        return out


class ResidualBlock2d(ResidualBlock3d):
    """2-D residual block (res_blocks.py:13-105): the same tree on 2-D convolutions."""

    _conv = Conv2d


class DepthwiseConv3d(torch.nn.Conv3d):
    """torch.nn.Conv3d(groups=in_channels, padding="same") on the HIP stencil kernel."""

    def forward(self, X):
        if (self.groups != self.in_channels or self.in_channels != self.out_channels
                or tuple(self.stride) != (1, 1, 1) or tuple(self.dilation) != (1, 1, 1)
                or self.padding != "same"):
            raise NotImplementedError("HIP DepthwiseConv3d: groups == channels, stride 1, "
                                      "padding='same'")
        return HF.dwconv3d(X, self.weight, self.bias)


class DepthwiseConv2d(torch.nn.Conv2d):
    """torch.nn.Conv2d(groups=in_channels, padding="same") as a depth-1 volume on the stencil
    kernel."""

    def forward(self, X):
        if (self.groups != self.in_channels or self.in_channels != self.out_channels
                or tuple(self.stride) != (1, 1) or tuple(self.dilation) != (1, 1)
                or self.padding != "same"):
            raise NotImplementedError("HIP DepthwiseConv2d: groups == channels, stride 1, "
                                      "padding='same'")
        return HF.dwconv3d(X.unsqueeze(2), self.weight.unsqueeze(2), self.bias).squeeze(2)


class DepthwiseConvNd(torch.nn.Module):
    """Mixin of the depthwise convolutions of the U-Net's ``conv_type="depthwise"`` blocks
    (unet.py:276-307: ``Conv(in, in, k, stride, padding, groups=in)``): the stencil kernel computes the
    stride-1 "same" convolution and the requested output is the lattice
    ``same[(k // 2 - p) + stride * o]`` of it -- a strided view, materialised by the consumer's
    NDHWC copy. Needs an odd kernel and padding <= k // 2 per axis (padding "same" included)."""

    def _lattice(self, same, k, stride, pad):
        index = [slice(None), slice(None)]
        for n, kk, st, p in zip(same.shape[2:], k, stride, pad):
            if kk % 2 == 0 or p > kk // 2:
                raise NotImplementedError("HIP depthwise conv: odd kernels, padding <= k // 2")
            out = (n + 2 * p - kk) // st + 1
            lo = kk // 2 - p
            index.append(slice(lo, lo + st * (out - 1) + 1, st))
        if all(sl == slice(0, n, 1) for sl, n in zip(index[2:], same.shape[2:])):
            return same
        return same[tuple(index)]


class DepthwiseConv3dStrided(torch.nn.Conv3d, DepthwiseConvNd):
    """torch.nn.Conv3d(c, c, k, stride, padding, groups=c) of the depthwise U-Net block."""

    def forward(self, X, X_cat=None):
        if self.groups != self.in_channels or self.in_channels != self.out_channels \
                or tuple(self.dilation) != (1, 1, 1):
            raise NotImplementedError("HIP DepthwiseConv3dStrided: groups == channels, dilation 1")
        if X_cat is not None:
            X = HF.cat_channels([X, X_cat])
        k = tuple(self.kernel_size)
        pad = tuple(kk // 2 for kk in k) if self.padding == "same" else tuple(self.padding)
        return self._lattice(HF.dwconv3d(X, self.weight, self.bias), k, tuple(self.stride), pad)


class DepthwiseConv2dStrided(torch.nn.Conv2d, DepthwiseConvNd):
    """The 2-D form on depth-1 volumes."""

    def forward(self, X, X_cat=None):
        if self.groups != self.in_channels or self.in_channels != self.out_channels \
                or tuple(self.dilation) != (1, 1):
            raise NotImplementedError("HIP DepthwiseConv2dStrided: groups == channels, dilation 1")
        if X_cat is not None:
            X = HF.cat_channels([X.unsqueeze(2), X_cat.unsqueeze(2)]).squeeze(2)
        k = tuple(self.kernel_size)
        pad = tuple(kk // 2 for kk in k) if self.padding == "same" else tuple(self.padding)
        same = HF.dwconv3d(X.unsqueeze(2), self.weight.unsqueeze(2), self.bias).squeeze(2)
        return self._lattice(same, k, tuple(self.stride), pad)


class ConvNeXtBlock3d(torch.nn.Module):
    """ConvNeXt block (adell_mri/modules/layers/res_blocks.py:516-604): depthwise conv ->
    LayerNorm over channels -> Linear -> GELU -> Linear -> layer scale -> + input
    (-> 1x1x1 conv + GELU when the channel count changes). Activations are NDHWC, so the
    reference's two permutes are no-ops here and every stage is a HIP kernel."""

    def __init__(self, in_channels: int, kernel_size: int, inter_channels: int,
                 out_channels: int, adn_fn: torch.nn.Module = torch.nn.Identity,
                 layer_scale_init_value: float = 1e-6, skip_activation: bool = None):
        super().__init__()
        self.in_channels = in_channels
        self.kernel_size = kernel_size
        self.inter_channels = inter_channels
        self.out_channels = out_channels
        self.adn_fn = adn_fn
        self.layer_scale_init_value = layer_scale_init_value
        self.skip_activation = skip_activation
        self.dwconv = self._depthwise(in_channels, in_channels, kernel_size=kernel_size,
                                      padding="same", groups=in_channels)
        self.norm = RowLayerNorm(in_channels, eps=1e-6)
        self.pwconv1 = Linear(in_channels, inter_channels)
        self.act = torch.nn.GELU()
        self.pwconv2 = Linear(inter_channels, in_channels)
        self.gamma = (torch.nn.Parameter(layer_scale_init_value * torch.ones((in_channels)),
                                         requires_grad=True)
                      if layer_scale_init_value > 0 else None)
        if out_channels != in_channels:
            self.out_layer = torch.nn.Sequential(
                self._pointwise(in_channels, out_channels, kernel_size=1, padding="same"),
                torch.nn.GELU())
        else:
            self.out_layer = None

    _depthwise, _pointwise = DepthwiseConv3d, Conv3d

    def forward(self, x):
        if x.dim() == 4:   # ConvNeXtBlock2d: the same kernels on a depth-1 volume
            return self._forward5(x.unsqueeze(2)).squeeze(2)
        return self._forward5(x)

    def _forward5(self, x):
        inp = ops.ndhwc(x)
        w = self.dwconv.weight
        h = HF.dwconv3d(inp, w if w.dim() == 5 else w.unsqueeze(2), self.dwconv.bias)
        rows = ops.ndhwc(h).permute(0, 2, 3, 4, 1)          # [N, D, H, W, C], contiguous
        rows = self.norm(rows)
        # pwconv2 -> layer scale -> + input as ONE GEMM: gamma * (h W^T + b) = h (gamma W)^T +
        # gamma b, so the scale is folded into the [C, 4C] weight (parameter algebra on a tiny
        # tensor; autograd carries dgamma / dW back through it) and the residual add rides the
        # GEMM epilogue. No full-size scale / add passes.
        if self.gamma is None:
            w2, b2 = self.pwconv2.weight, self.pwconv2.bias
        else:
            w2, b2 = HF.rowscale(self.gamma, self.pwconv2.weight, self.pwconv2.bias)
        res = inp.permute(0, 2, 3, 4, 1)
        if HF.mlp_ok(rows, self.pwconv1.weight, w2):
            # GELU inside the epilogues of pwconv1 (forward) and of the dY W2 GEMM (backward): no
            # element-wise pass over the 4 C-wide hidden layer
            rows = HF.mlp(rows, self.pwconv1.weight, self.pwconv1.bias, w2, b2, act="gelu",
                          residual=res)
        else:
            rows = HF.linear(HF.elementwise(self.pwconv1(rows), act="gelu"), w2, b2, residual=res)
        out = rows.permute(0, 4, 1, 2, 3)
        if self.out_layer is not None:
            proj = self.out_layer[0]
            y = proj(out.squeeze(2)).unsqueeze(2) if proj.weight.dim() == 4 else proj(out)
            out = HF.norm_drop_act(y, act="gelu")
        return out


class ConvNeXtBlock2d(ConvNeXtBlock3d):
    """Two-dimensional ConvNeXt block (res_blocks.py:429-513): same arithmetic and kernels as the
    3-D block on depth-1 volumes; parameters keep their 2-D shapes (``dwconv.weight`` [C, 1, k, k],
    ``out_layer.0.weight`` [Co, Ci, 1, 1])."""

    _depthwise, _pointwise = DepthwiseConv2d, Conv2d
