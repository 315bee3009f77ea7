"""Loss, fused optimiser and whole training-step parity on the GPU (through the C
ABI), against the oracle and the reference-generated golden fixtures."""
import os

import numpy as np
import pytest
import torch

from adell_mri_amd import ops
from adell_mri_amd.modules.activations import activation_factory
from adell_mri_amd.modules.segmentation.losses import (CompoundLoss, binary_focal_loss,
                                                       binary_generalized_dice_loss)
from adell_mri_amd.modules.segmentation.pl import UNetPL
from adell_mri_amd.optim import FusedAdamW, FusedSGD
from adell_mri_amd.trainer import StepRunner
from cases import UNET_CASES, grad_rel_err
from oracle import cops
from oracle.weights import tensor_for

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def yaml_loss():
    return CompoundLoss([(binary_generalized_dice_loss, {"smooth": 1e-5, "eps": 1e-6}),
                         (binary_focal_loss, {"gamma": 1.0, "eps": 1e-6})])


@pytest.mark.parametrize("gamma", [1.0, 2.0, 1.5])
@pytest.mark.parametrize("shape", [(2, 1, 8, 8, 8), (1, 1, 33, 17, 9), (3, 1, 40, 40, 24)])
def test_dice_focal_forward_backward_match_oracle(cuda, gamma, shape):
    rng = np.random.default_rng(0)
    p = rng.uniform(0, 1, shape).astype(np.float32)
    p.flat[:3] = [0.0, 1.0, 1e-7]  # clamp branches
    t = (rng.uniform(0, 1, shape) > 0.8).astype(np.float32)
    B = shape[0]
    d_ref, f_ref, g_ref = cops.dice_focal(p, t, 1e-5, 1e-6, gamma, 1e-6, grad=True,
                                          gscale_dice=0.5 / B, gscale_focal=0.5 / B)
    pt = torch.from_numpy(p).to(cuda).requires_grad_(True)
    tt = torch.from_numpy(t).to(cuda)
    loss_fn = CompoundLoss([(binary_generalized_dice_loss, {"smooth": 1e-5, "eps": 1e-6}),
                            (binary_focal_loss, {"gamma": gamma, "eps": 1e-6})])
    d, f = loss_fn(pt, tt)
    np.testing.assert_allclose(d.detach().cpu().numpy(), d_ref, rtol=2e-5)
    np.testing.assert_allclose(f.detach().cpu().numpy(), f_ref, rtol=2e-5)
    torch.stack([d.mean(), f.mean()]).mean().backward()
    np.testing.assert_allclose(pt.grad.cpu().numpy(), g_ref, rtol=2e-4, atol=1e-9)


def test_golden_loss_values(cuda):
    g = np.load(os.path.join(GOLD, "blocks.npz"))
    p, t = torch.from_numpy(g["loss_p"]).to(cuda), torch.from_numpy(g["loss_t"]).to(cuda)
    d = binary_generalized_dice_loss(p, t, smooth=1e-5, eps=1e-6)
    f = binary_focal_loss(p, t, gamma=1.0, eps=1e-6)
    f2 = binary_focal_loss(p, t, gamma=2.0, eps=1e-6)
    np.testing.assert_allclose(d.cpu().numpy(), g["loss_dice"], rtol=1e-5)
    np.testing.assert_allclose(f.cpu().numpy(), g["loss_focal"].reshape(-1), rtol=1e-5)
    np.testing.assert_allclose(f2.cpu().numpy(), g["loss_focal_g2"].reshape(-1), rtol=1e-5)


def test_fused_sgd_matches_oracle_and_torch(cuda):
    rng = np.random.default_rng(1)
    shapes = [(7, 3, 3, 3, 3), (5,), (64, 32, 3, 3, 3), (1,)]
    ps = [torch.nn.Parameter(torch.from_numpy(rng.standard_normal(s).astype(np.float32)).to(cuda))
          for s in shapes]
    ref_p = [p.detach().cpu().numpy().copy() for p in ps]
    ref_b = [np.zeros_like(r) for r in ref_p]
    opt = FusedSGD(ps, lr=5e-4, momentum=0.99, weight_decay=5e-3, nesterov=True)
    for step in range(3):
        opt.zero_grad(set_to_none=(step % 2 == 0))   # both conventions
        grads = [rng.standard_normal(s).astype(np.float32) for s in shapes]
        for p, g in zip(ps, grads):
            if p.grad is None:
                p.grad = torch.from_numpy(g).to(cuda)
            else:
                p.grad.copy_(torch.from_numpy(g))
        opt.step()
        for r, g, b in zip(ref_p, grads, ref_b):
            cops.sgd_nesterov(r, g, b, 5e-4, 0.99, 5e-3, True, first=(step == 0))
    for p, r in zip(ps, ref_p):
        np.testing.assert_allclose(p.detach().cpu().numpy(), r, rtol=1e-6, atol=1e-7)


def test_fused_adamw_matches_torch(cuda):
    rng = np.random.default_rng(2)
    w0 = rng.standard_normal((33, 17)).astype(np.float32)
    a = torch.nn.Parameter(torch.from_numpy(w0.copy()).to(cuda))
    b = torch.nn.Parameter(torch.from_numpy(w0.copy()))
    oa = FusedAdamW([a], lr=5e-3, weight_decay=1e-3)
    ob = torch.optim.AdamW([b], lr=5e-3, weight_decay=1e-3)
    for _ in range(4):
        g = rng.standard_normal(w0.shape).astype(np.float32)
        oa.zero_grad()
        a.grad = torch.from_numpy(g).to(cuda)
        b.grad = torch.from_numpy(g.copy())
        oa.step()
        ob.step()
    np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().numpy(), rtol=2e-5, atol=1e-6)


def test_fused_adam_matches_torch(cuda):
    """torch.optim.Adam ("adam" of optimizer_factory.py:5-14): decay coupled into the gradient."""
    from adell_mri_amd.modules.segmentation.pl import get_optimizer
    rng = np.random.default_rng(3)
    w0 = rng.standard_normal((29, 13)).astype(np.float32)
    a = torch.nn.Parameter(torch.from_numpy(w0.copy()).to(cuda))
    b = torch.nn.Parameter(torch.from_numpy(w0.copy()))
    oa = get_optimizer("adam", [a], lr=5e-3, weight_decay=1e-2)
    ob = torch.optim.Adam([b], lr=5e-3, weight_decay=1e-2)
    for _ in range(4):
        g = rng.standard_normal(w0.shape).astype(np.float32)
        oa.zero_grad()
        a.grad = torch.from_numpy(g).to(cuda)
        b.grad = torch.from_numpy(g.copy())
        oa.step()
        ob.step()
    np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().numpy(), rtol=2e-5, atol=1e-6)


def test_ema_update(cuda):
    s = torch.randn(1000, device=cuda)
    p = torch.randn(1000, device=cuda)
    ref = s - (1 - 0.99) * (s - p)
    ops.ema_update(s, p, 0.99)
    assert torch.allclose(s, ref, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("name", ["unet3d_cfg2_small", "unet3d_cfg2_tiny"])
def test_training_step_matches_reference_after_one_sgd_step(cuda, name):
    """training_step arithmetic (pl.py:284-317,382-421) + SGD-Nesterov (pl.py:563-569):
    loss, gradients and post-step parameters against the real reference."""
    g = np.load(os.path.join(GOLD, name + ".npz"))
    kw = dict(UNET_CASES[name])
    kw["activation_fn"] = activation_factory[kw["activation_fn"]]
    kw["dropout_param"] = 0.0  # the fixture was produced with dropout disabled (eval)
    net = UNetPL(image_key="image", label_key="mask", learning_rate=5e-4, weight_decay=5e-3,
                 loss_fn=yaml_loss(), **kw)
    net.load_state_dict({k: torch.from_numpy(tensor_for(k, v.shape))
                         for k, v in net.state_dict().items()})
    net = net.to(cuda).train()
    runner = StepRunner(net)
    batch = {"image": torch.from_numpy(g["x"]).to(cuda), "mask": torch.from_numpy(g["y"]).to(cuda)}
    loss = runner.train_step(batch)
    np.testing.assert_allclose(float(loss.detach()), g["loss"], rtol=1e-4)
    for k, p in net.named_parameters():
        err = grad_rel_err(g, k, p.grad.cpu().numpy())
        assert err < 2e-3, (k, err)
        np.testing.assert_allclose(p.detach().cpu().numpy(), g["step1:" + k], rtol=1e-4, atol=2e-7)
    # a second step must see the updated weights (packed-weight cache invalidation)
    loss2 = runner.train_step(batch)
    assert float(loss2) != float(loss)
