"""BASELINE configs[0]: the 2-D U-Net of the reference's testing/test_unet.py:63-72 with the
constructor defaults (BatchNorm2d with batch statistics, PReLU incl. its weight gradient;
dropout off: torch's mask stream is not reproducible;
transposed-conv upscaling, strided-conv encoder) against a fixture generated from the real
reference (oracle/make_golden.py, case unet2d_cfg1)."""
import os

import numpy as np
import pytest
import torch

from adell_mri_amd.modules.segmentation.unet import UNet
from cases import grad_rel_err
from oracle.torch_ref.unet import compound_loss
from oracle.weights import fill_state_dict

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KW = dict(spatial_dimensions=2, depth=[16, 32, 64], upscale_type="transpose", padding="same",
          strides=[2, 2, 2], kernel_sizes=[3, 3, 3], conv_type="regular", link_type="identity",
          activation_fn=torch.nn.PReLU, dropout_param=0.0)


def build():
    net = UNet(**KW)
    net.load_state_dict(fill_state_dict(net.state_dict()))
    return net


def test_unet2d_parameters_equal_reference():
    g = np.load(os.path.join(GOLD, "unet2d_cfg1.npz"))
    net = build()
    assert sum(p.numel() for p in net.parameters()) == 140748      # SURVEY.md 8(d), cfg 1
    assert [k for k, _ in net.named_parameters()] == [str(k) for k in g["param_keys"]]


@pytest.mark.gpu
def test_unet2d_logits_grads_and_sgd_step_match_reference(cuda):
    from adell_mri_amd.optim import FusedSGD

    g = np.load(os.path.join(GOLD, "unet2d_cfg1.npz"))
    net = build().to(cuda).train()          # batch statistics, as in the fixture
    x, y = torch.from_numpy(g["x"]).to(cuda), torch.from_numpy(g["y"]).to(cuda)
    logits, _ = net(x, return_logits=True)
    ref = g["logits"]
    assert np.abs(logits.detach().cpu().numpy() - ref).max() / np.abs(ref).max() < 1e-4
    opt = FusedSGD(net.parameters(), lr=5e-4, momentum=0.99, weight_decay=5e-3, nesterov=True)
    opt.zero_grad()
    prob, _ = net(x)
    loss = compound_loss(prob, y)
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-4)
    loss.backward()
    for k, p in net.named_parameters():
        assert p.grad is not None, k
        assert grad_rel_err(g, k, p.grad.cpu().numpy()) < 3e-3, k
    opt.step()
    for k, p in net.named_parameters():
        np.testing.assert_allclose(p.detach().cpu().numpy(), g["step1:" + k], rtol=1e-4, atol=1e-6,
                                   err_msg=k)


@pytest.mark.gpu
@pytest.mark.parametrize("per_channel", [False, True])
@pytest.mark.parametrize("drop_p", [0.0, 0.3])
def test_prelu_weight_gradient(cuda, per_channel, drop_p):
    """d/da of norm -> dropout -> PReLU against autograd on the same dropout mask."""
    from adell_mri_amd import functional as HF

    g = torch.Generator().manual_seed(3)
    C = 12
    x = torch.randn((2, C, 5, 6, 7), generator=g).to(cuda)
    a = (0.25 + 0.1 * torch.randn((C if per_channel else 1,), generator=g)).to(cuda)
    a.requires_grad_(True)
    torch.manual_seed(5)
    y = HF.norm_drop_act(x, norm="instance", act="prelu", act_w=a, drop_p=drop_p, training=True)
    r = torch.randn(y.shape, generator=g).to(cuda)
    (y * r).sum().backward()
    # reference: recover u = dropout(norm(x)) from the same kernel with an identity activation
    # is not possible (new mask), so use y itself: u = y where y >= 0 else y / a
    with torch.no_grad():
        aw = a.detach().view(1, -1, 1, 1, 1)
        u = torch.where(y >= 0, y, y / aw)
        contrib = torch.where(u < 0, r * u, torch.zeros_like(u))
        want = contrib.sum(dim=(0, 2, 3, 4)) if per_channel else contrib.sum().view(1)
    assert torch.allclose(a.grad, want, rtol=2e-4, atol=1e-5), (a.grad, want)
