"""Convolution leaves. They subclass the stock ``torch.nn`` modules so that
parameter shapes, initialisation and ``state_dict`` keys are those of the
reference's ``torch.nn.Conv3d`` / ``ConvTranspose3d`` call sites
(adell_mri/modules/segmentation/unet.py:260-273, :445-458), but ``forward`` runs
the MI355X kernels and never ``F.conv3d``."""
import torch

from ... import functional as HF
from ..._lib import AdellHipError


def _resolve_padding(padding, kernel_size, stride, dilation):
    if isinstance(padding, str):
        if padding == "valid":
            return (0,) * len(kernel_size)
        if padding == "same":
            if any(k % 2 == 0 for k in kernel_size):
                raise AdellHipError("padding='same' needs odd kernel sizes on the HIP path")
            return tuple(d * (k - 1) // 2 for k, d in zip(kernel_size, dilation))
        raise ValueError(padding)
    return tuple(padding)


class Conv3d(torch.nn.Conv3d):
    def _dilated(self):
        """kernel 3, stride 1, padding "same", dilation > 1 (the atrous pyramid's convs)."""
        return (any(d != 1 for d in self.dilation) and tuple(self.kernel_size) == (3, 3, 3)
                and tuple(self.stride) == (1, 1, 1) and self.padding == "same")

    def _check(self):
        if self.groups != 1 or self.padding_mode != "zeros" or (
                any(d != 1 for d in self.dilation) and not self._dilated()):
            raise AdellHipError("HIP Conv3d supports groups=1, zero padding, dilation=1 (or kernel 3, "
                                "stride 1, padding='same' with a dilation)")

    def takes_carry(self):
        """Whether forward() hands GradCarry arguments to the conv kernels (the space-to-depth
        stem path does not)."""
        k, st = tuple(self.kernel_size), tuple(self.stride)
        pad = _resolve_padding(self.padding, self.kernel_size, self.stride, self.dilation)
        return not (k == st and max(k) > 2 and tuple(pad) == (0, 0, 0)) and not self._dilated()

    def rows_spec(self):
        """(weight, stride, padding) for functional.expect_rows when this conv may read its input
        as split rows (3x3x3, stride 1: the layers that carry the FLOPs of a U-Net), else None."""
        if (self.groups != 1 or any(d != 1 for d in self.dilation) or self.padding_mode != "zeros"
                or tuple(self.kernel_size) != (3, 3, 3) or tuple(self.stride) != (1, 1, 1)):
            return None
        pad = _resolve_padding(self.padding, self.kernel_size, self.stride, self.dilation)
        return (self.weight, (1, 1, 1), tuple(pad))

    def forward(self, X, X_cat=None, residual=None, carry_in=None, carry_out=None,
                carry_x0=None, carry_cat=None):
        """``X_cat``: second source of a virtual channel concat; ``residual``: tensor
        added to the output inside the kernel epilogue; ``carry_in`` / ``carry_out`` /
        ``carry_x0`` / ``carry_cat``: functional.GradCarry (residual link, skip fork)."""
        self._check()
        if self._dilated():
            if X_cat is not None or residual is not None or carry_in is not None \
                    or carry_out is not None or carry_x0 is not None or carry_cat is not None:
                raise AdellHipError("HIP Conv3d: a dilated conv takes one source and no fused epilogue")
            return HF.conv3d_dilated(X, self.weight, self.bias, self.dilation)
        pad = _resolve_padding(self.padding, self.kernel_size, self.stride, self.dilation)
        k, st = tuple(self.kernel_size), tuple(self.stride)
        if k == st and max(k) > 2 and tuple(pad) == (0, 0, 0) and X_cat is None:
            return self._patchify(X, residual)
        return HF.conv3d(X, self.weight, self.bias, self.stride, pad, x1=X_cat, residual=residual,
                         carry_in=carry_in, carry_out=carry_out, carry_x0=carry_x0,
                         carry_cat=carry_cat)

    def _patchify(self, X, residual):
        """kernel == stride, no padding (ViT / ConvNeXt stems): space-to-depth view of the
        input, then a 1x1x1 convolution over K^3*Cin channels on the MFMA kernel."""
        N, C, D, H, W = X.shape
        kd, kh, kw = self.kernel_size
        Do, Ho, Wo = D // kd, H // kh, W // kw
        X = X[:, :, :Do * kd, :Ho * kh, :Wo * kw]
        Xp = (X.reshape(N, C, Do, kd, Ho, kh, Wo, kw).permute(0, 2, 4, 6, 3, 5, 7, 1)
              .reshape(N, Do, Ho, Wo, kd * kh * kw * C).permute(0, 4, 1, 2, 3))
        w = self.weight.permute(0, 2, 3, 4, 1).reshape(self.out_channels, -1, 1, 1, 1)
        return HF.conv3d(Xp, w, self.bias, 1, 0, residual=residual)


class Conv2d(torch.nn.Conv2d):
    """2-D convolution run as a depth-1 3-D convolution."""

    def forward(self, X, X_cat=None, residual=None):
        if self.groups != 1 or any(d != 1 for d in self.dilation) or self.padding_mode != "zeros":
            raise AdellHipError("HIP Conv2d supports groups=1, dilation=1, zero padding only")
        pad = _resolve_padding(self.padding, self.kernel_size, self.stride, self.dilation)
        un = lambda t: None if t is None else t.unsqueeze(2)  # noqa: E731
        y = HF.conv3d(un(X), self.weight.unsqueeze(2), self.bias, (1, *self.stride), (0, *pad),
                      x1=un(X_cat), residual=un(residual))
        part = getattr(y, "_adell_partials", None)
        y = y.squeeze(2)
        if part is not None:
            y._adell_partials = part
        return y


class ConvTranspose3d(torch.nn.ConvTranspose3d):
    def forward(self, X, output_size=None):
        k, s = tuple(self.kernel_size), tuple(self.stride)
        ok = (k == s and all(f in (1, 2) for f in k) and tuple(self.padding) == (0, 0, 0)
              and tuple(self.output_padding) == (0, 0, 0) and self.groups == 1
              and tuple(self.dilation) == (1, 1, 1) and output_size is None)
        if not ok:
            raise AdellHipError(
                "HIP ConvTranspose3d implements kernel == stride in {1,2} per dim, padding 0 (the "
                f"U-Net decoder upscaling); got k={self.kernel_size} s={self.stride} "
                f"p={self.padding}")
        return HF.conv_transpose3d(X, self.weight, self.bias)


class ConvTranspose2d(torch.nn.ConvTranspose2d):
    """2-D transposed convolution (kernel == stride) run as a depth-1 3-D one."""

    def forward(self, X, output_size=None):
        k, s = tuple(self.kernel_size), tuple(self.stride)
        ok = (k == s and all(f in (1, 2) for f in k) and tuple(self.padding) == (0, 0)
              and tuple(self.output_padding) == (0, 0) and self.groups == 1
              and tuple(self.dilation) == (1, 1) and output_size is None)
        if not ok:
            raise AdellHipError("HIP ConvTranspose2d implements kernel == stride in {1,2}, padding 0; "
                                f"got k={self.kernel_size} s={self.stride} p={self.padding}")
        return HF.conv_transpose3d(X.unsqueeze(2), self.weight.unsqueeze(2), self.bias).squeeze(2)


class Upsample(torch.nn.Upsample):
    """torch.nn.Upsample for the "upsample" upscaling path (unet.py:419-443): linear modes
    ("bilinear" on 4-D, "trilinear" on 5-D input, align_corners False) and "nearest"."""

    def forward(self, X):
        if self.size is not None or self.align_corners or self.recompute_scale_factor:
            raise AdellHipError("HIP Upsample implements scale_factor with align_corners=False")
        if self.mode == "nearest":
            sf = self.scale_factor
            sf = (sf,) * (X.dim() - 2) if not isinstance(sf, (tuple, list)) else tuple(sf)
            size = [int(n * f) for n, f in zip(X.shape[2:], sf)]
            if X.dim() == 4:
                return HF.interpolate_nearest(X.unsqueeze(2), [1] + size).squeeze(2)
            return HF.interpolate_nearest(X, size)
        want = "bilinear" if X.dim() == 4 else "trilinear"
        if self.mode != want:   # torch raises for a mode / rank mismatch as well
            raise NotImplementedError(f"Got {X.dim()}D input, but {self.mode} mode needs "
                                      f"{'4' if self.mode == 'bilinear' else '5'}D input")
        return HF.upsample_linear(X, self.scale_factor)


class MaxPool3d(torch.nn.MaxPool3d):
    def forward(self, X):
        if self.ceil_mode or self.return_indices or ops_triple(self.dilation) != (1, 1, 1):
            raise AdellHipError("HIP MaxPool3d supports ceil_mode=False, dilation=1 only")
        return HF.max_pool3d(X, self.kernel_size, self.stride, self.padding)


class MaxPool2d(torch.nn.MaxPool2d):
    """2-D max pooling run as a depth-1 3-D one."""

    def forward(self, X):
        pair = lambda v: (v, v) if isinstance(v, int) else tuple(v)  # noqa: E731
        if self.ceil_mode or self.return_indices or pair(self.dilation) != (1, 1):
            raise AdellHipError("HIP MaxPool2d supports ceil_mode=False, dilation=1 only")
        k, st, p = pair(self.kernel_size), pair(self.stride), pair(self.padding)
        return HF.max_pool3d(X.unsqueeze(2), (1, *k), (1, *st), (0, *p)).squeeze(2)


def ops_triple(v):
    return (v,) * 3 if isinstance(v, int) else tuple(v)
