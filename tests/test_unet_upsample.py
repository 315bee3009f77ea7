"""Constructor corners of the U-Net: 2-D residual skip links (ResidualBlock2d,
res_blocks.py:13-105) and upscale_type="upsample" (the reference's constructor default: 1x1 conv + torch.nn.Upsample,
unet.py:419-443) against fixtures generated from the real reference (oracle/make_golden.py cases
unet2d_upsample: all-default 2-D U-Net; unet3d_upsample: trilinear, anisotropic last stride), and
the interpolation kernels against torch."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from adell_mri_amd.modules.activations import activation_factory
from adell_mri_amd.modules.segmentation.unet import UNet
from cases import grad_rel_err
from oracle.torch_ref.unet import compound_loss
from oracle.weights import fill_state_dict

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = {
    "unet2d_upsample": dict(spatial_dimensions=2, depth=[8, 16, 32], padding="same",
                            strides=[2, 2, 2], kernel_sizes=[3, 3, 3],
                            activation_fn=torch.nn.PReLU, dropout_param=0.0),
    "unet2d_residual_links": dict(spatial_dimensions=2, depth=[8, 16, 32], padding=1,
                                  strides=[2, 2, 2], kernel_sizes=[3, 3, 3],
                                  upscale_type="transpose", norm_type="instance",
                                  activation_fn=activation_factory["swish"], dropout_param=0.0,
                                  link_type="residual", in_channels=2),
    "unet2d_resnet_blocks": dict(spatial_dimensions=2, depth=[8, 16, 32], padding=1,
                                 strides=[2, 2, 2], kernel_sizes=[3, 3, 3], conv_type="resnet",
                                 upscale_type="transpose", norm_type="instance",
                                 activation_fn=activation_factory["swish"], dropout_param=0.0,
                                 link_type="identity", in_channels=1),
    "unet3d_upsample": dict(spatial_dimensions=3, depth=[8, 16, 32], padding=1,
                            strides=[2, 2, [2, 2, 1]], kernel_sizes=[3, 3, 3],
                            upscale_type="upsample", interpolation="trilinear",
                            norm_type="instance", activation_fn=activation_factory["swish"],
                            dropout_param=0.0, link_type="identity", in_channels=2),
}


def build(name):
    net = UNet(**CASES[name])
    net.load_state_dict(fill_state_dict(net.state_dict()))
    return net


@pytest.mark.parametrize("name", list(CASES))
def test_parameter_names_equal_reference(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    assert [k for k, _ in build(name).named_parameters()] == [str(k) for k in g["param_keys"]]


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CASES))
def test_logits_and_gradients_match_reference(cuda, name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    net = build(name).to(cuda).train()
    x, y = torch.from_numpy(g["x"]).to(cuda), torch.from_numpy(g["y"]).to(cuda)
    logits, _ = net(x, return_logits=True)
    ref = g["logits"]
    assert np.abs(logits.detach().cpu().numpy() - ref).max() / np.abs(ref).max() < 1e-4
    prob, _ = net(x)
    loss = compound_loss(prob, y)
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-4)
    loss.backward()
    for k, p in net.named_parameters():
        assert p.grad is not None, k
        assert grad_rel_err(g, k, p.grad.cpu().numpy()) < 3e-3, k


@pytest.mark.gpu
@pytest.mark.parametrize("shape,scale", [((2, 5, 4, 6, 7), (2, 2, 2)), ((1, 3, 3, 5, 4), (2, 2, 1)),
                                         ((1, 4, 2, 3, 5), (3, 1.5, 2)), ((2, 6, 9, 7), (2, 2))])
def test_linear_upsample_matches_torch(cuda, shape, scale):
    from adell_mri_amd import functional as HF
    g = torch.Generator().manual_seed(len(shape) + int(scale[0]))
    x = torch.randn(shape, generator=g, dtype=torch.float64).requires_grad_(True)
    mode = "bilinear" if len(shape) == 4 else "trilinear"
    y = F.interpolate(x, scale_factor=scale, mode=mode, align_corners=False)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(dy)
    xd = x.detach().float().to(cuda).requires_grad_(True)
    yd = HF.upsample_linear(xd, scale)
    assert tuple(yd.shape) == tuple(y.shape)
    assert float((yd.detach().cpu().double() - y.detach()).abs().max()) < 1e-5
    yd.backward(dy.float().to(cuda))
    assert float((xd.grad.cpu().double() - x.grad).abs().max()) < 1e-4
