"""The n_classes > 2 softmax head and the bottleneck-classifier path on the HIP kernels against
stock torch (reference ops: torch.nn.Softmax(dim=1) unet.py:641-655, X.flatten(2).max(-1) and
torch.nn.Linear unet.py:691-695, 826-828)."""
import pytest
import torch

from adell_mri_amd import functional as HF
from adell_mri_amd import ops
from adell_mri_amd.modules.activations import activation_factory
from adell_mri_amd.modules.segmentation.unet import UNet

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(2, 3, 5, 6, 7), (1, 5, 8, 8, 8), (2, 12, 3, 4, 5), (3, 4, 9, 11)])
def test_channel_softmax_fwd_bwd(cuda, shape):
    g = torch.Generator().manual_seed(shape[1])
    x = (torch.randn(shape, generator=g) * 3).requires_grad_(True)
    r = torch.randn(shape, generator=g)
    y = torch.softmax(x, 1)
    (y * r).sum().backward()
    xd = x.detach().to(cuda).requires_grad_(True)
    yd = HF.channel_softmax(xd if len(shape) == 4 else ops.ndhwc(xd))
    (yd * r.to(cuda)).sum().backward()
    assert torch.allclose(yd.detach().cpu(), y.detach(), rtol=1e-5, atol=1e-7)
    assert torch.allclose(xd.grad.cpu(), x.grad, rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("shape", [(2, 40, 4, 5, 6), (1, 300, 3, 3, 3), (2, 8, 6, 7)])
def test_channel_max_fwd_bwd(cuda, shape):
    g = torch.Generator().manual_seed(shape[1])
    x = torch.randn(shape, generator=g).requires_grad_(True)
    r = torch.randn(shape[:2], generator=g)
    p = x.flatten(start_dim=2).max(-1).values
    (p * r).sum().backward()
    xd = x.detach().to(cuda).requires_grad_(True)
    pd = HF.channel_max(xd if len(shape) == 4 else ops.ndhwc(xd))
    (pd * r.to(cuda)).sum().backward()
    assert torch.equal(pd.detach().cpu(), p.detach())
    assert torch.equal(xd.grad.cpu().reshape(shape), x.grad)


def test_multiclass_unet_with_bottleneck_classifier_runs_on_hip(cuda):
    """n_classes = 4: probabilities sum to one over the class axis; classifier output shape;
    every parameter receives a gradient through the HIP softmax / max / GEMM path."""
    net = UNet(spatial_dimensions=3, conv_type="regular", link_type="identity",
               upscale_type="transpose", norm_type="instance", padding=1, dropout_param=0.0,
               activation_fn=activation_factory["swish"], in_channels=1, n_classes=4,
               depth=[8, 16], kernel_sizes=[3, 3], strides=[2, 2],
               bottleneck_classification=True).to(cuda).train()
    assert type(net.bottleneck_classifier).__module__.startswith("adell_mri_amd")
    x = torch.rand(2, 1, 16, 16, 16, device=cuda)
    prob, bn = net(x)
    assert prob.shape == (2, 4, 16, 16, 16) and bn.shape == (2, 4)
    assert torch.allclose(prob.sum(1), torch.ones_like(prob.sum(1)), atol=1e-5)
    logits = net(x, return_logits=True)[0]
    assert torch.allclose(torch.softmax(logits, 1), prob, atol=1e-6)
    (prob[:, 1].mean() + bn.pow(2).mean()).backward()
    missing = [k for k, p in net.named_parameters() if p.grad is None]
    assert not missing, missing
