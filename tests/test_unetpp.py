"""U-Net++ mirror: CPU key parity and GPU parity against the reference fixture."""
import os

import numpy as np
import pytest
import torch

from adell_mri_amd import functional as HF
from adell_mri_amd.modules.activations import activation_factory
from adell_mri_amd.modules.segmentation.unetpp import UNetPlusPlus
from cases import UNETPP_CASES, grad_rel_err
from oracle.torch_ref.unet import compound_loss
from oracle.weights import tensor_for

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


NAMES = ["unetpp3d_small", "unetpp2d_small"]


def build(name="unetpp3d_small"):
    kw = dict(UNETPP_CASES[name])
    kw["activation_fn"] = activation_factory[kw["activation_fn"]]
    net = UNetPlusPlus(**kw)
    net.load_state_dict({k: torch.from_numpy(tensor_for(k, v.shape))
                         for k, v in net.state_dict().items()})
    return net


@pytest.mark.parametrize("name", NAMES)
def test_unetpp_state_dict_keys_and_shapes_equal_reference(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    sd = build(name).state_dict()
    assert list(sd.keys()) == [str(k) for k in g["param_keys"]]
    shapes = {str(k): tuple(int(i) for i in str(s).split(",")) for k, s in
              zip(g["param_keys"], g["param_shapes"])}
    for k, v in sd.items():
        assert tuple(v.shape) == shapes[k], k


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_unetpp_logits_aux_and_grads_match_reference(cuda, name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    net = build(name).to(cuda).eval()
    x = torch.from_numpy(g["x"]).to(cuda)
    with torch.no_grad():
        logits = net(x, return_logits=True)[0]
    ref = g["logits"]
    assert np.abs(logits.cpu().numpy() - ref).max() / np.abs(ref).max() < 1e-4
    prob, bn, aux = net(x)
    assert bn is None and len(aux) == len(UNETPP_CASES[name]["depth"]) - 2
    for i, a in enumerate(aux):
        np.testing.assert_allclose(a.detach().cpu().numpy(), g[f"aux{i}"], rtol=1e-4, atol=1e-5)
    loss = compound_loss(prob, torch.from_numpy(g["y"]).to(cuda))
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-4)
    loss.backward()
    for k, p in net.named_parameters():
        if ("grad:" + k) not in g.files:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        assert grad_rel_err(g, k, p.grad.cpu().numpy()) < 2e-3, k


@pytest.mark.gpu
@pytest.mark.parametrize("insz,outsz", [((4, 4, 4), (8, 8, 8)), ((3, 5, 2), (7, 5, 9)),
                                         ((8, 8, 8), (4, 4, 4))])
def test_nearest_resample_and_concat_match_torch(cuda, insz, outsz):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 6, *insz, generator=g).requires_grad_(True)
    ref = torch.nn.functional.interpolate(x, outsz)
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy)
    xd = x.detach().to(cuda).requires_grad_(True)
    y = HF.interpolate_nearest(xd, outsz)
    y.backward(dy.to(cuda))
    assert torch.equal(y.detach().cpu(), ref.detach())
    assert torch.allclose(xd.grad.cpu(), x.grad, rtol=1e-6, atol=1e-6)
    parts = [torch.randn(2, c, *insz, generator=g) for c in (4, 3, 8)]
    pd = [p.to(cuda).requires_grad_(True) for p in parts]
    cat = HF.cat_channels(pd)
    assert torch.equal(cat.detach().cpu(), torch.cat(parts, 1))
    w = torch.randn(cat.shape, generator=g)
    (cat * w.to(cuda)).sum().backward()
    off = 0
    for p, q in zip(pd, parts):
        assert torch.equal(p.grad.cpu(), w[:, off:off + q.shape[1]])
        off += q.shape[1]


@pytest.mark.gpu
@pytest.mark.parametrize("size,k,s,p", [((8, 8, 8), 2, 2, 1), ((9, 7, 5), 3, 2, 1),
                                        ((8, 8, 8), 2, 2, 0), ((6, 6, 5), (2, 2, 1), (2, 2, 1), (1, 1, 0)),
                                        ((5, 5, 5), 3, 1, 1)])
@pytest.mark.parametrize("C,ties", [(6, False), (8, False), (8, True)])
def test_maxpool3d_fwd_bwd_matches_torch(cuda, size, k, s, p, C, ties):
    """(C = 8 with kernel == stride: the 16-bytes-of-channels row kernels; ties: integer-valued
    inputs, the first maximum in scan order takes the gradient as in torch)"""
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, C, *size, generator=g)
    if ties:
        x = torch.round(x)
    x.requires_grad_(True)
    ref = torch.nn.functional.max_pool3d(x, k, s, p)
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy)
    xd = x.detach().to(cuda).requires_grad_(True)
    y = HF.max_pool3d(xd, k, s, p)
    y.backward(dy.to(cuda))
    assert torch.equal(y.detach().cpu(), ref.detach())
    assert torch.allclose(xd.grad.cpu(), x.grad, rtol=1e-6, atol=1e-6)
