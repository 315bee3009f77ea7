"""Model-level parity of the HIP U-Net with the reference (golden fixtures made
by oracle/make_golden.py from the real adell_mri code). north_star tolerance:
logits within 1e-4 relative of the reference PyTorch-CPU forward."""
import os

import numpy as np
import pytest
import torch

from adell_mri_amd.modules.activations import activation_factory
from adell_mri_amd.modules.segmentation.unet import UNet
from cases import ASP_CASES, DEPTHWISE_CASES, SAE_CASES, UNET_CASES, grad_rel_err
from oracle.torch_ref.unet import compound_loss
from oracle.weights import tensor_for

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def build(kw, device):
    kw = dict(kw)
    kw["activation_fn"] = activation_factory[kw["activation_fn"]]
    net = UNet(**kw)
    sd = {k: torch.from_numpy(tensor_for(k, v.shape)) for k, v in net.state_dict().items()}
    net.load_state_dict(sd)
    return net.to(device)


@pytest.fixture(params=["f16x3", "fp32"])
def precision(request):
    """Both MFMA modes of the convolutions must meet the parity bar."""
    from adell_mri_amd import functional as HF

    old = HF.CONV_PRECISION
    HF.set_conv_precision(request.param)
    yield request.param
    HF.set_conv_precision(old)


@pytest.mark.parametrize("name", list(UNET_CASES))
def test_logits_within_1e4_of_reference(cuda, name, precision):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    net = build(UNET_CASES[name], cuda).eval()
    x = torch.from_numpy(g["x"]).to(cuda)
    with torch.no_grad():
        logits, bn = net(x, return_logits=True)
        prob, _ = net(x)
    assert bn is None
    ref = g["logits"]
    got = logits.cpu().numpy()
    assert got.shape == ref.shape
    rel = np.abs(got - ref).max() / np.abs(ref).max()
    print(f"{name} [{precision}] logits rel err {rel:.2e}")
    assert rel < 1e-4, rel
    np.testing.assert_allclose(prob.cpu().numpy(), g["prob"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("name", list(UNET_CASES))
def test_parameter_gradients_match_reference(cuda, name, precision):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    net = build(UNET_CASES[name], cuda).eval()  # eval: dropout off, as in the fixture
    x = torch.from_numpy(g["x"]).to(cuda)
    y = torch.from_numpy(g["y"]).to(cuda)
    prob, _ = net(x)
    loss = compound_loss(prob, y)
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-4)
    loss.backward()
    worst = 0.0
    for k, p in net.named_parameters():
        assert p.grad is not None, k
        err = grad_rel_err(g, k, p.grad.cpu().numpy())
        worst = max(worst, err)
        assert err < 2e-3, (k, err)
    print("worst relative grad error", worst)


def test_train_mode_dropout_runs_and_is_seeded(cuda):
    net = build(UNET_CASES["unet3d_cfg2_small"], cuda).train()
    x = torch.rand(2, 2, 16, 16, 16, device=cuda)
    torch.manual_seed(3)
    a, _ = net(x, return_logits=True)
    b, _ = net(x, return_logits=True)
    assert torch.isfinite(a).all() and not torch.equal(a, b)
    a.sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())


# ---- conv_type="depthwise" (unet.py:276-307): depthwise stencil kernels + 1x1 convs, including the
# reference's padded 1x1 conv of every downsampling block (its output grows by 2 voxels per axis and
# the decoder crops the skip tensors) --------------------------------------------------------------
@pytest.mark.parametrize("name", list(DEPTHWISE_CASES))
def test_depthwise_unet_matches_reference(cuda, name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    net = build(DEPTHWISE_CASES[name], cuda).eval()
    x = torch.from_numpy(g["x"]).to(cuda)
    y = torch.from_numpy(g["y"]).to(cuda)
    with torch.no_grad():
        logits, _ = net(x, return_logits=True)
    ref = g["logits"]
    assert tuple(logits.shape) == ref.shape
    rel = np.abs(logits.cpu().numpy() - ref).max() / np.abs(ref).max()
    assert rel < 1e-4, rel
    prob, _ = net(x)
    loss = compound_loss(prob, y)
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-4)
    loss.backward()
    # a 1x1 conv on ONE input channel in front of an instance norm (the 2-D case's first block) has a
    # mathematically zero weight gradient too: the scale of a key is floored by 1e-3 of the largest
    # gradient of the network
    gmax = max(float(np.abs(g[f]).max()) for f in g.files if f.startswith("grad:"))
    for k, p in net.named_parameters():
        assert p.grad is not None, k
        ref, got = g["grad:" + k], p.grad.cpu().numpy()
        if float(np.abs(ref).max()) < 1e-3 * gmax:
            # (both sides hold rounding noise of a zero gradient: an absolute bar)
            assert float(np.abs(got - ref).max()) < 1e-4 * gmax, k
        else:
            assert grad_rel_err(g, k, got) < 2e-3, k


# ---- conv_type="sae" (unet.py:375-397): every conv block followed by a concurrent (spatial + channel)
# squeeze-and-excite, 3-D and 2-D, against fixtures from the real reference classes ------------------
# conv_type="asp": atrous-pyramid encoder ops (dilated convs on the interleaved sub-lattices of their
# input, functional.conv3d_dilated; odd extents 9 / 5 exercise the zero frame), "sae" decoder ops
@pytest.mark.parametrize("name", list(SAE_CASES) + list(ASP_CASES))
def test_sae_unet_matches_reference(cuda, name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    net = build({**SAE_CASES, **ASP_CASES}[name], cuda).eval()
    assert [k for k, _ in net.named_parameters()] == [f[5:] for f in g.files if f.startswith("grad:")]
    x = torch.from_numpy(g["x"]).to(cuda)
    y = torch.from_numpy(g["y"]).to(cuda)
    with torch.no_grad():
        logits, _ = net(x, return_logits=True)
    ref = g["logits"]
    assert tuple(logits.shape) == ref.shape
    rel = np.abs(logits.cpu().numpy() - ref).max() / np.abs(ref).max()
    assert rel < 1e-4, rel
    prob, _ = net(x)
    loss = compound_loss(prob, y)
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-4)
    loss.backward()
    gmax = max(float(np.abs(g[f]).max()) for f in g.files if f.startswith("grad:"))
    for k, p in net.named_parameters():
        assert p.grad is not None, k
        ref, got = g["grad:" + k], p.grad.cpu().numpy()
        scale = max(float(np.abs(ref).max()), 1e-3 * gmax)
        assert float(np.abs(got - ref).max()) < 2e-3 * scale, k
