"""BASELINE config-2 sizes (2-channel 128^3 volumes, 32 / 64-channel 128^3 feature maps), where
no CPU oracle finishes in seconds: size-independent properties of the convolution path.

  * linearity in the input:      conv(a x1 + b x2) = a conv(x1) + b conv(x2)      (bias off)
  * adjointness (dot-product test): <conv_W(x), y> = <x, conv_bwd_data_W(y)> = <W, conv_bwd_weight(x, y)>
    -- ties the three kernels (specialised igemm forward, backward-data, z-ring backward-weight)
    to one another at the size the benchmark runs them
  * the statistics the conv epilogue emits = the statistics of the tensor it wrote
  * the full config-2 U-Net at 128^3: probabilities in (0, 1), bit-identical across two runs in
    eval mode (every reduction is fixed-order), instance-normalised activations have mean 0 / var 1
"""
import pytest
import torch

from adell_mri_amd import functional as HF
from adell_mri_amd import ops

pytestmark = pytest.mark.gpu
S = 128


def _dot(a, b):
    return float((a.double() * b.double()).sum())


@pytest.mark.parametrize("cin,cout", [(32, 32), (64, 64)])
def test_conv_linearity_and_adjointness_at_128_cubed(cuda, cin, cout):
    g = torch.Generator(device="cuda").manual_seed(cin)
    x1 = ops.ndhwc(torch.randn((1, cin, S, S, S), generator=g, device=cuda))
    x2 = ops.ndhwc(torch.randn((1, cin, S, S, S), generator=g, device=cuda))
    w = torch.randn((cout, cin, 3, 3, 3), generator=g, device=cuda) * 0.05
    wp, wpb = ops.pack_weight_f16x3(w, 0), ops.pack_weight_f16x3(w, 1)

    def conv(x):
        return ops.conv3d_fwd(x, wp, None, cout, 3, 1, 1)[0]

    y1, y2 = conv(x1), conv(x2)
    y12 = conv(ops.ndhwc(0.75 * x1 - 1.5 * x2))
    lin = 0.75 * y1 - 1.5 * y2
    assert float((y12 - lin).abs().max() / lin.abs().max()) < 2e-6
    del y12, lin, x2, y2
    # adjoint identities with a random cotangent
    dy = ops.ndhwc(torch.randn((1, cout, S, S, S), generator=g, device=cuda))
    lhs = _dot(y1, dy)
    dx, _ = ops.conv3d_bwd_data(dy, wpb, (S, S, S), cin, 0, 3, 1, 1)
    assert abs(_dot(x1, dx) - lhs) < 1e-5 * abs(lhs) + 1e-3 * float(dy.numel()) ** 0.5
    del dx
    dw = ops.conv3d_bwd_weight(x1, dy, 3, 1, 1, f16x3=True)
    assert abs(_dot(w, dw) - lhs) < 1e-5 * abs(lhs) + 1e-3 * float(dy.numel()) ** 0.5


def test_epilogue_statistics_match_the_written_tensor_at_128_cubed(cuda):
    g = torch.Generator(device="cuda").manual_seed(1)
    x = ops.ndhwc(torch.randn((2, 32, S, S, S), generator=g, device=cuda))
    w = torch.randn((32, 32, 3, 3, 3), generator=g, device=cuda) * 0.05
    b = torch.randn((32,), generator=g, device=cuda)
    y, part = ops.conv3d_fwd(x, ops.pack_weight_f16x3(w, 0), b, 32, 3, 1, 1, want_stats=True)
    tot = part.double().sum(1)                                   # [N, C, 2]
    yd = y.double()
    want = torch.stack([yd.sum((2, 3, 4)), (yd * yd).sum((2, 3, 4))], -1)
    assert float((tot - want).abs().max() / want.abs().max()) < 1e-6
    z = HF.norm_drop_act(y, norm="instance", act="identity")
    zd = z.double()
    assert float(zd.mean((2, 3, 4)).abs().max()) < 1e-4
    assert float((zd.var((2, 3, 4), unbiased=False) - 1).abs().max()) < 1e-3


def test_config2_unet_forward_at_full_size_is_deterministic(cuda):
    from adell_mri_amd.modules.activations import activation_factory
    from adell_mri_amd.modules.segmentation.unet import UNet
    torch.manual_seed(0)
    net = UNet(spatial_dimensions=3, conv_type="regular", link_type="residual",
               upscale_type="transpose", norm_type="instance", padding=1, dropout_param=0.1,
               activation_fn=activation_factory["swish"], in_channels=2, n_classes=2,
               depth=[32, 32, 64, 128, 256], kernel_sizes=[3] * 5, strides=[2] * 5).to(cuda).eval()
    assert sum(p.numel() for p in net.parameters()) == 8264303     # = the reference (8.26 M)
    x = torch.rand((2, 2, S, S, S), device=cuda)
    with torch.no_grad():
        p1, _ = net(x)
        p2, _ = net(x)
    assert tuple(p1.shape) == (2, 1, S, S, S)
    assert torch.equal(p1, p2)
    assert float(p1.min()) > 0.0 and float(p1.max()) < 1.0 and bool(torch.isfinite(p1).all())
    # batch items are independent under instance norm: item 0 alone gives the same answer
    with torch.no_grad():
        q, _ = net(x[:1].contiguous())
    assert float((q - p1[:1]).abs().max()) < 1e-6


def test_stride2_downsampling_layer_full_size_fused_vs_generic_kernels(cuda, monkeypatch):
    """The three one-launch kernels of the 32 -> 32 stride-2 layer (conv_fwd_s2 / conv_dgrad_s2 /
    conv_wgrad_s2.hip) at the benchmark's level-0 size (2 x 128^3 in, 2 x 64^3 out) against the
    implicit-GEMM / parity-class / generic weight-gradient kernels they replace (different
    decompositions of the same sums: agreement to f16x3 accuracy), plus the statistics partials."""
    import torch

    from adell_mri_amd import _lib
    from adell_mri_amd import functional as HF
    from adell_mri_amd import ops

    g = torch.Generator().manual_seed(11)
    x = ops.ndhwc(torch.randn(2, 32, 128, 128, 128, generator=g).to(cuda))
    w = (torch.randn(32, 32, 3, 3, 3, generator=g) * 0.05).to(cuda)
    b = torch.randn(32, generator=g).to(cuda)
    dy = ops.ndhwc((torch.randn(2, 32, 64, 64, 64, generator=g) * 1e-3).to(cuda))

    def run():
        xg = x.detach().requires_grad_(True)
        wg, bg = w.detach().requires_grad_(True), b.detach().requires_grad_(True)
        y = HF.conv3d(xg, wg, bg, stride=2, padding=1, want_stats=True)
        part = y._adell_partials.double().sum(1)
        y.backward(dy)
        return y.detach(), part, xg.grad, wg.grad, bg.grad

    fused = run()
    monkeypatch.setitem(ops.FLAGS, "no_s2fused", True)
    L = _lib.lib()
    L.adell_set_tuning(b"wgrad_nozring", 1)     # also turns the sub-lattice weight-gradient kernel off
    try:
        generic = run()
    finally:
        L.adell_set_tuning(b"wgrad_nozring", 0)
    for name, a_, b_ in zip(("y", "partials", "dx", "dw", "db"), fused, generic):
        rel = float((a_ - b_).abs().max()) / float(b_.abs().max())
        assert rel < 5e-6, (name, rel)
