"""ResNet-backbone U-Net (BASELINE config 2b in miniature: 7^3 stem, k=5 / k=3
bottleneck residual stages with BatchNorm, MaxPool with anisotropic [2,2,1] strides,
anisotropic transposed convs, crop_to_size re-alignment) against the reference."""
import os

import numpy as np
import pytest
import torch

from adell_mri_amd import functional as HF
from adell_mri_amd.modules.activations import activation_factory
from adell_mri_amd.modules.layers.adn_fn import get_adn_fn
from adell_mri_amd.modules.layers.res_net import ResNet, resnet_to_encoding_ops
from adell_mri_amd.modules.segmentation.unet import UNet
from cases import grad_rel_err
from oracle import cops
from oracle.torch_ref.unet import compound_loss
from oracle.weights import tensor_for

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
STRUCTURE = [[8, 8, 5, 2], [16, 16, 3, 2]]
MAXPOOL = [[2, 2, 1], [2, 2, 2]]


def build():
    res = ResNet(dict(spatial_dim=3, in_channels=2, structure=STRUCTURE, maxpool_structure=MAXPOOL,
                      res_type="resnet", adn_fn=get_adn_fn(3, "batch", "swish", 0.0)))
    enc = resnet_to_encoding_ops([res])[0]
    net = UNet(spatial_dimensions=3, encoding_operations=enc, upscale_type="transpose",
               link_type="identity", norm_type="instance", padding=1, dropout_param=0.0,
               activation_fn=activation_factory["swish"], in_channels=2, n_classes=2,
               depth=[STRUCTURE[0][0], *[x[0] for x in STRUCTURE]], kernel_sizes=[3, 3, 3],
               strides=[2, *MAXPOOL])
    sd = net.state_dict()
    net.load_state_dict({k: (torch.from_numpy(tensor_for(k, v.shape))
                             if v.is_floating_point() and "running_" not in k else v)
                         for k, v in sd.items()})
    return net


def test_backbone_unet_state_dict_keys_equal_reference():
    g = np.load(os.path.join(GOLD, "unet3d_resnet_backbone.npz"))
    net = build()
    assert [k for k, _ in net.named_parameters()] == [str(k) for k in g["param_keys"]]


@pytest.mark.gpu
def test_backbone_unet_logits_and_grads_match_reference(cuda):
    g = np.load(os.path.join(GOLD, "unet3d_resnet_backbone.npz"))
    net = build().to(cuda).train()  # batch statistics, as in the fixture
    x = torch.from_numpy(g["x"]).to(cuda)
    logits, _ = net(x, return_logits=True)
    ref = g["logits"]
    rel = np.abs(logits.detach().cpu().numpy() - ref).max() / np.abs(ref).max()
    assert rel < 1e-4, rel
    prob, _ = net(x)
    loss = compound_loss(prob, torch.from_numpy(g["y"]).to(cuda))
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-4)
    loss.backward()
    for k, p in net.named_parameters():
        assert grad_rel_err(g, k, p.grad.cpu().numpy()) < 3e-3, k


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["f16x3"])
@pytest.mark.parametrize("N,Cin,size,Cout,k,s,p", [
    (1, 2, (12, 12, 12), 16, 7, 1, 3), (1, 16, (9, 9, 9), 16, 5, 1, 2),
    (1, 64, (8, 8, 8), 64, 5, 1, 2)])
def test_big_kernel_convs_fwd_bwd(cuda, prec, N, Cin, size, Cout, k, s, p):
    from adell_mri_amd import ops

    rng = np.random.default_rng(5)
    x = rng.standard_normal((N, Cin, *size)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, k, k, k)) / np.sqrt(Cin * k ** 3)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    ref = cops.conv3d(x, w, b, s, p)
    dy = rng.standard_normal(ref.shape).astype(np.float32)
    dx_ref, dw_ref, db_ref = cops.conv3d_bwd(x, w, dy, s, p)
    old = HF.CONV_PRECISION
    HF.set_conv_precision(prec)
    try:
        xd = ops.ndhwc(torch.from_numpy(x).to(cuda)).requires_grad_(True)
        wd = torch.from_numpy(w).to(cuda).requires_grad_(True)
        bd = torch.from_numpy(b).to(cuda).requires_grad_(True)
        y = HF.conv3d(xd, wd, bd, s, p)
        y.backward(ops.ndhwc(torch.from_numpy(dy).to(cuda)))
    finally:
        HF.set_conv_precision(old)
    rel = lambda a, r: float(np.abs(a - r).max() / (np.abs(r).max() + 1e-30))  # noqa: E731
    assert rel(y.detach().cpu().numpy(), ref) < 1e-5
    assert rel(xd.grad.cpu().numpy(), dx_ref) < 1e-5
    assert rel(wd.grad.cpu().numpy(), dw_ref) < 1e-5
    assert rel(bd.grad.cpu().numpy(), db_ref) < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("factors", [(2, 2, 1), (1, 2, 2), (2, 2, 2)])
def test_anisotropic_transposed_conv(cuda, factors):
    rng = np.random.default_rng(6)
    x = rng.standard_normal((2, 16, 4, 5, 3)).astype(np.float32)
    w = (rng.standard_normal((16, 8, *factors)) * 0.2).astype(np.float32)
    b = rng.standard_normal(8).astype(np.float32)
    ref = cops.conv_transpose3d(x, w, b, factors, 0)
    dy = rng.standard_normal(ref.shape).astype(np.float32)
    dx_ref, dw_ref, db_ref = cops.conv_transpose3d_bwd(x, w, dy, factors, 0)
    xd = torch.from_numpy(x).to(cuda).requires_grad_(True)
    wd = torch.from_numpy(w).to(cuda).requires_grad_(True)
    bd = torch.from_numpy(b).to(cuda).requires_grad_(True)
    y = HF.conv_transpose3d(xd, wd, bd)
    y.backward(torch.from_numpy(dy).to(cuda))
    rel = lambda a, r: float(np.abs(a - r).max() / (np.abs(r).max() + 1e-30))  # noqa: E731
    assert rel(y.detach().cpu().numpy(), ref) < 1e-5
    assert rel(xd.grad.cpu().numpy(), dx_ref) < 1e-5
    assert rel(wd.grad.cpu().numpy(), dw_ref) < 1e-5
    assert rel(bd.grad.cpu().numpy(), db_ref) < 2e-5


@pytest.mark.gpu
def test_patchify_conv_module_matches_oracle(cuda):
    """kernel == stride == 4 (ConvNeXt / ViT stems) through the space-to-depth path."""
    from adell_mri_amd.modules.layers.conv import Conv3d

    rng = np.random.default_rng(8)
    x = rng.standard_normal((2, 3, 16, 16, 8)).astype(np.float32)
    m = Conv3d(3, 24, 4, stride=4)
    ref = cops.conv3d(x, m.weight.detach().numpy(), m.bias.detach().numpy(), 4, 0)
    dy = rng.standard_normal(ref.shape).astype(np.float32)
    dx_ref, dw_ref, db_ref = cops.conv3d_bwd(x, m.weight.detach().numpy(), dy, 4, 0)
    m = m.to(cuda)
    xd = torch.from_numpy(x).to(cuda).requires_grad_(True)
    y = m(xd)
    y.backward(torch.from_numpy(dy).to(cuda))
    rel = lambda a, r: float(np.abs(a - r).max() / (np.abs(r).max() + 1e-30))  # noqa: E731
    assert rel(y.detach().cpu().numpy(), ref) < 1e-5
    assert rel(xd.grad.cpu().numpy(), dx_ref) < 1e-5
    assert rel(m.weight.grad.cpu().numpy(), dw_ref) < 1e-5
    assert rel(m.bias.grad.cpu().numpy(), db_ref) < 2e-5
