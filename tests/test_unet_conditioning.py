"""Tabular feature conditioning of the U-Net skip connections (unet.py:716-740, 803-810) against
a fixture generated from the real reference (oracle/make_golden.py, case unet3d_feature_cond), the
per-(item, channel) scale kernel against torch, and U-out (regularization.py:11-57)."""
import os

import numpy as np
import pytest
import torch

from adell_mri_amd.modules.activations import activation_factory
from adell_mri_amd.modules.segmentation.unet import UNet
from cases import grad_rel_err
from oracle.torch_ref.unet import compound_loss
from oracle.weights import fill_state_dict

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KW = dict(spatial_dimensions=3, depth=[8, 16, 32], padding=1, strides=[2, 2, 2],
          kernel_sizes=[3, 3, 3], upscale_type="transpose", norm_type="instance",
          activation_fn=activation_factory["swish"], dropout_param=0.0, link_type="identity",
          in_channels=1, feature_conditioning=5)


def build():
    net = UNet(**KW)
    net.load_state_dict(fill_state_dict(net.state_dict()))
    return net


def test_parameter_names_equal_reference():
    g = np.load(os.path.join(GOLD, "unet3d_feature_cond.npz"))
    assert [k for k, _ in build().named_parameters()] == [str(k) for k in g["param_keys"]]


@pytest.mark.gpu
def test_feature_conditioned_unet_matches_reference(cuda):
    g = np.load(os.path.join(GOLD, "unet3d_feature_cond.npz"))
    net = build().to(cuda).train()      # batch statistics in the BatchNorm1d gates
    x, y = torch.from_numpy(g["x"]).to(cuda), torch.from_numpy(g["y"]).to(cuda)
    fc = torch.from_numpy(g["x_fc"]).to(cuda)
    logits, _ = net(x, X_feature_conditioning=fc, return_logits=True)
    ref = g["logits"]
    assert np.abs(logits.detach().cpu().numpy() - ref).max() / np.abs(ref).max() < 1e-4
    prob, _ = net(x, X_feature_conditioning=fc)
    loss = compound_loss(prob, y)
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-4)
    loss.backward()
    for k, p in net.named_parameters():
        if not p.requires_grad:
            continue
        assert p.grad is not None, k
        assert grad_rel_err(g, k, p.grad.cpu().numpy()) < 3e-3, k


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(3, 6, 5, 7, 9), (2, 40, 4, 4), (4, 16)])
def test_scale_per_item_channel_matches_torch(cuda, shape):
    from adell_mri_amd import functional as HF
    g = torch.Generator().manual_seed(len(shape))
    x = torch.randn(shape, generator=g, dtype=torch.float64).requires_grad_(True)
    s = torch.randn(shape[:2], generator=g, dtype=torch.float64).requires_grad_(True)
    y = x * s.view(*shape[:2], *([1] * (len(shape) - 2)))
    dy = torch.randn(shape, generator=g, dtype=torch.float64)
    y.backward(dy)
    xd = x.detach().float().to(cuda).requires_grad_(True)
    sd = s.detach().float().to(cuda).requires_grad_(True)
    yd = HF.scale_per_item_channel(xd, sd)
    assert float((yd.detach().cpu().double() - y.detach()).abs().max()) < 1e-5
    yd.backward(dy.float().to(cuda))
    assert float((xd.grad.cpu().double() - x.grad).abs().max()) < 1e-5
    assert float((sd.grad.cpu().double() - s.grad).abs().max()) < 1e-4 * max(1.0, float(s.grad.abs().max()))


@pytest.mark.gpu
def test_uout_scales_every_item_and_channel_uniformly(cuda):
    """X' = X (1 + r), r ~ U(-beta, beta) constant over the spatial axes; identity in eval mode;
    inside an ADN (norm -> U-out -> activation) the network still trains."""
    from adell_mri_amd.modules.layers.regularization import UOut
    x = torch.rand((8, 16, 4, 6, 5), generator=torch.Generator().manual_seed(0)).to(cuda) + 0.5
    m = UOut(0.1).train()
    y = m(x)
    ratio = (y / x).flatten(2)
    assert float((ratio.max(-1).values - ratio.min(-1).values).max()) < 1e-5   # one r per (n, c)
    r = ratio[..., 0] - 1.0
    assert float(r.abs().max()) <= 0.1 + 1e-6 and float(r.std()) > 0.03
    assert torch.equal(m.eval()(x), x)
    net = UNet(spatial_dimensions=3, depth=[4, 8], padding=1, strides=[2, 2], kernel_sizes=[3, 3],
               upscale_type="transpose", norm_type="instance", dropout_type="uout",
               dropout_param=0.1, activation_fn=activation_factory["swish"], in_channels=1).to(cuda)
    net.train()
    out, _ = net(torch.rand((2, 1, 8, 8, 8), device=cuda))
    out.sum().backward()
    assert all(torch.isfinite(p.grad).all() for p in net.parameters() if p.grad is not None)


KW_SKIP = dict(spatial_dimensions=3, depth=[8, 16, 32], padding=1, strides=[2, 2, 2],
               kernel_sizes=[3, 3, 3], upscale_type="transpose", norm_type="instance",
               activation_fn=activation_factory["swish"], dropout_param=0.0, link_type="conv",
               in_channels=1, skip_conditioning=1, deep_supervision=True)


@pytest.mark.gpu
def test_skip_conditioning_and_deep_supervision_match_reference(cuda):
    """X_skip_layer resized (nearest) to every skip resolution and concatenated (unet.py:796-801),
    deep-supervision heads (unet.py:657-683, 836-841): logits, every auxiliary output and every
    gradient of the main loss against the reference fixture."""
    g = np.load(os.path.join(GOLD, "unet3d_skipcond_deepsup.npz"))
    net = UNet(**KW_SKIP)
    net.load_state_dict(fill_state_dict(net.state_dict()))
    assert [k for k, _ in net.named_parameters()] == [str(k) for k in g["param_keys"]]
    net = net.to(cuda).eval()
    x, y = torch.from_numpy(g["x"]).to(cuda), torch.from_numpy(g["y"]).to(cuda)
    sk = torch.from_numpy(g["x_skip"]).to(cuda)
    logits = net(x, X_skip_layer=sk, return_logits=True)[0]
    ref = g["logits"]
    assert np.abs(logits.detach().cpu().numpy() - ref).max() / np.abs(ref).max() < 1e-4
    prob, _, deep = net(x, X_skip_layer=sk)
    for i, o in enumerate(deep):
        np.testing.assert_allclose(o.detach().cpu().numpy(), g[f"aux{i}"], rtol=1e-4, atol=1e-5)
    loss = compound_loss(prob, y)
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-4)
    loss.backward()
    for k, p in net.named_parameters():
        if ("grad:" + k) in g.files:   # the auxiliary heads do not enter the main loss
            assert grad_rel_err(g, k, p.grad.cpu().numpy()) < 3e-3, k


@pytest.mark.gpu
@pytest.mark.parametrize("shape,size", [((2, 1, 12, 10, 8), (5, 7, 3)), ((1, 3, 9, 11), (4, 6)),
                                        ((2, 2, 6, 6, 6), (1, 13, 6))])
def test_aligned_linear_resize_matches_torch(cuda, shape, size):
    """F.interpolate(size=..., align_corners=True) of the deep-supervision targets (pl.py:305-309)."""
    import torch.nn.functional as F

    from adell_mri_amd import functional as HF
    x = torch.rand(shape, generator=torch.Generator().manual_seed(7))
    mode = "bilinear" if len(shape) == 4 else "trilinear"
    want = F.interpolate(x, size, mode=mode, align_corners=True)
    got = HF.resize_linear_aligned(x.to(cuda), size).cpu()
    assert tuple(got.shape) == tuple(want.shape)
    assert float((got - want).abs().max()) < 1e-5


@pytest.mark.gpu
def test_feature_conditioned_unetr_matches_reference(cuda):
    """Tabular feature gates on the UNETR skip tensors (unetr.py:217-218, 356-358, 399-405)
    against the fixture unetr3d_feature_cond made from the real reference."""
    from adell_mri_amd.modules.segmentation.unetr import UNETR
    from cases import UNETR_CASES

    g = np.load(os.path.join(GOLD, "unetr3d_feature_cond.npz"))
    kw = dict(UNETR_CASES["unetr3d_feature_cond"])
    kw["activation_fn"] = activation_factory[kw["activation_fn"]]
    net = UNETR(**kw)
    assert [k for k, _ in net.named_parameters()] == [str(k) for k in g["param_keys"]]
    net.load_state_dict(fill_state_dict(net.state_dict()))
    net = net.to(cuda).eval()       # the gates' BatchNorm1d layers on their running statistics
    x, y = torch.from_numpy(g["x"]).to(cuda), torch.from_numpy(g["y"]).to(cuda)
    fc = torch.from_numpy(g["x_fc"]).to(cuda)
    logits, _ = net(x, X_feature_conditioning=fc, return_logits=True)
    ref = g["logits"]
    assert np.abs(logits.detach().cpu().numpy() - ref).max() / np.abs(ref).max() < 1e-4
    prob, _ = net(x, X_feature_conditioning=fc)
    loss = compound_loss(prob, y)
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-4)
    loss.backward()
    for k, p in net.named_parameters():
        if not p.requires_grad or ("grad:" + k) not in g.files:
            continue
        assert p.grad is not None, k
        assert grad_rel_err(g, k, p.grad.cpu().numpy()) < 3e-3, k
