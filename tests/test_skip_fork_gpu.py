"""U-Net skip forks (unet.py:768-822): a level output feeds the downsampling conv and, later, the
link op / the decoder's concat. The later reader's gradient is parked in a functional.GradCarry
and added in the epilogue of the downsampling conv's backward-data kernel (plain kernel:
adell_conv3d_bwd_data_f16x3_add, parity classes: adell_conv3d_bwd_data_s2_f16x3_add) instead of by
autograd's full-size accumulation pass. Same gradients either way."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)


@pytest.mark.parametrize("n,cin,cout,size,pad", [(2, 32, 32, (16, 16, 16), 1),
                                                 (1, 16, 48, (8, 12, 20), 1),
                                                 (1, 64, 32, (8, 8, 8), 0)])
def test_parity_class_backward_data_adds_a_dx_shaped_operand(cuda, n, cin, cout, size, pad):
    from adell_mri_amd import functional as HF
    from adell_mri_amd import ops

    g = torch.Generator().manual_seed(cin)
    w = (torch.randn(cout, cin, 3, 3, 3, generator=g) * 0.05).to(cuda)
    osz = ops.conv_out_size(size, (3,) * 3, (2,) * 3, (pad,) * 3)
    dy = ops.ndhwc(torch.randn(n, cout, *osz, generator=g).to(cuda))
    add0 = ops.ndhwc(torch.randn(n, cin, *size, generator=g).to(cuda))
    classes = HF._packed_s2_classes(w, (pad,) * 3)
    plain = ops.conv3d_bwd_data_s2(dy, classes, size, cin, (pad,) * 3)
    fused = ops.conv3d_bwd_data_s2(dy, classes, size, cin, (pad,) * 3, add0=add0)
    want = plain + add0
    # the add happens on the fp32 accumulator before the store: one rounding apart at most
    assert _rel(fused, want) <= 2e-7
    assert fused.shape == want.shape


def _unet(link_type, cuda, depth=(16, 16, 32)):
    from adell_mri_amd.modules.layers.adn_fn import activation_factory
    from adell_mri_amd.modules.segmentation.unet import UNet

    torch.manual_seed(3)
    return UNet(spatial_dimensions=3, conv_type="regular", link_type=link_type,
                upscale_type="transpose", norm_type="instance", padding=1, dropout_param=0.0,
                activation_fn=activation_factory["swish"], in_channels=2, n_classes=2,
                depth=list(depth), kernel_sizes=[3] * len(depth),
                strides=[2] * len(depth)).to(cuda).train()


def _step(net, x, r):
    net.zero_grad()
    xg = x.clone().requires_grad_(True)
    y = net(xg)[0]
    (y * r).sum().backward()
    return y.detach(), xg.grad.clone(), {k: p.grad.clone() for k, p in net.named_parameters()}


@pytest.mark.parametrize("link_type", ["residual", "conv", "identity"])
@pytest.mark.parametrize("classes", [False, True])
def test_fork_gradient_rides_the_downsampling_conv(cuda, monkeypatch, link_type, classes):
    from adell_mri_amd import functional as HF
    from adell_mri_amd import ops

    monkeypatch.setitem(HF.FLAGS, "s2class_always", classes)   # both backward-data kernels
    net = _unet(link_type, cuda)
    x = torch.randn(2, 2, 32, 32, 32, device=cuda)
    r = torch.randn(2, 1, 32, 32, 32, device=cuda)
    taken = []
    take = HF.GradCarry.take

    def spy(self):
        g = take(self)
        taken.append(g is not None)
        return g

    monkeypatch.setattr(HF.GradCarry, "take", spy)
    y_f, gx_f, gw_f = _step(net, x, r)
    # the two forks (levels 0 and 1) delivered a gradient to the downsampling convs (residual
    # links: plus one carry per block)
    assert sum(taken) >= 2, taken
    monkeypatch.setitem(ops.FLAGS, "no_skip_fork", True)
    taken.clear()
    y_a, gx_a, gw_a = _step(net, x, r)
    # (the forward is the same computation either way, but a level output with ONE reader may be
    # handed over as split rows and with two as fp32: a conv that reads rows and one that reads the
    # fp32 tensor agree to an ulp, not to the bit -- tools/dbg_rows16.py)
    assert _rel(y_f, y_a) <= 1e-6
    assert _rel(gx_f, gx_a) <= 2e-5
    # (a conv bias in front of an instance norm has a mathematically zero gradient: what is left is
    # rounding noise of the forward, floored by the largest gradient of the net)
    top = max(float(g.abs().max()) for g in gw_a.values())
    for k, g in gw_a.items():
        err = float((gw_f[k] - g).abs().max())
        assert err <= max(2e-5 * float(g.abs().max()), 1e-6 * top), k


def test_fork_is_off_under_no_grad(cuda):
    """No fork under no_grad or when the input needs no gradient (the first level output then
    still does, through the weights); the net runs and every parameter gets a gradient."""
    net = _unet("residual", cuda)
    x = torch.randn(1, 2, 16, 16, 16, device=cuda)
    with torch.no_grad():
        y0 = net(x)[0]
    y1 = net(x)[0]
    assert torch.equal(y0, y1)
    y1.sum().backward()
    assert all(p.grad is not None for p in net.parameters())


def test_frozen_downsampling_conv_still_delivers_the_fork(cuda):
    """The downsampling conv's weights frozen: its backward-data still runs (its input needs a
    gradient) and still adds the parked gradient."""
    from adell_mri_amd import ops

    net = _unet("residual", cuda)
    for p in net.encoding_operations[0][1].parameters():
        p.requires_grad_(False)
    x = torch.randn(2, 2, 16, 16, 16, device=cuda)
    r = torch.randn(2, 1, 16, 16, 16, device=cuda)
    _, gx_f, gw_f = _step_trainable(net, x, r)
    ops.FLAGS["no_skip_fork"] = True
    try:
        _, gx_a, gw_a = _step_trainable(net, x, r)
    finally:
        ops.FLAGS["no_skip_fork"] = False
    assert _rel(gx_f, gx_a) <= 2e-5
    # (a conv bias in front of an instance norm has a mathematically zero gradient: what is left is
    # rounding noise of the forward, floored by the largest gradient of the net)
    top = max(float(g.abs().max()) for g in gw_a.values())
    for k, g in gw_a.items():
        err = float((gw_f[k] - g).abs().max())
        assert err <= max(2e-5 * float(g.abs().max()), 1e-6 * top), k


def _step_trainable(net, x, r):
    net.zero_grad()
    xg = x.clone().requires_grad_(True)
    y = net(xg)[0]
    (y * r).sum().backward()
    return (y.detach(), xg.grad.clone(),
            {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None})


def test_hook_on_a_level_output_sees_the_whole_gradient(cuda, monkeypatch):
    """a tensor hook on a forked level output must see the SUM of its readers'
    gradients. A watched tensor (hooks / retain_grad) gets no carry: autograd accumulates."""
    from adell_mri_amd import ops

    net = _unet("residual", cuda)
    x = torch.randn(1, 2, 32, 32, 32, device=cuda)
    r = torch.randn(1, 1, 32, 32, 32, device=cuda)
    seen = {}

    def watch(name):
        def fwd_hook(_mod, _inp, out):
            out.register_hook(lambda g: seen.__setitem__(name, g.detach().clone()))
        return fwd_hook

    handles = [net.encoding_operations[i][0].register_forward_hook(watch(i)) for i in (0, 1)]
    _, gx_h, gw_h = _step(net, x, r)
    with_hooks = dict(seen)
    seen.clear()
    monkeypatch.setitem(ops.FLAGS, "no_grad_carry", True)       # plain autograd everywhere
    _, gx_p, gw_p = _step(net, x, r)
    for i in (0, 1):
        assert _rel(with_hooks[i], seen[i]) <= 1e-6
    for h in handles:
        h.remove()
    assert _rel(gx_h, gx_p) <= 1e-5
    for k in gw_p:
        assert _rel(gw_h[k], gw_p[k]) <= 1e-4, k
