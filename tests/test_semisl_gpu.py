"""Semi-supervised U-Net (-m gpu): the fused local contrastive loss against the fixture generated
from the reference's LocalContrastiveLoss (values, both gradients) and against the CPU oracle at a
larger size; UNetSemiSL features and the UNetContrastiveSemiSL.training_step arithmetic against the
fixtures from the reference's UNetSemiSL (semi_supervised_segmentation/pl.py:244-281, 371-450)."""
import os

import numpy as np
import pytest
import torch

from adell_mri_amd import functional as HF
from adell_mri_amd.modules.activations import activation_factory
from adell_mri_amd.modules.segmentation.losses import (CompoundLoss, binary_focal_loss,
                                                       binary_generalized_dice_loss)
from adell_mri_amd.modules.semi_supervised_segmentation import (LocalContrastiveLoss,
                                                                UNetContrastiveSemiSL, UNetSemiSL)
from adell_mri_amd.trainer import StepRunner
from adell_mri_amd.utils.utils import ExponentialMovingAverage
from cases import SEMISL_CASES
from oracle.torch_ref.semi_sl import local_contrastive_loss
from oracle.weights import tensor_for

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _close(a, ref, rel, what):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    err = np.abs(a - ref).max()
    assert err <= rel * np.abs(ref).max() + 1e-9, (what, err, np.abs(ref).max())


@pytest.mark.parametrize("tag", list("abcd"))
def test_loco_loss_matches_reference_fixture(cuda, tag):
    g = np.load(os.path.join(GOLD, "loco_loss.npz"))
    x1 = torch.from_numpy(g[f"{tag}:x1"]).to(cuda).requires_grad_(True)
    x2 = torch.from_numpy(g[f"{tag}:x2"]).to(cuda).requires_grad_(True)
    val = LocalContrastiveLoss(temperature=float(g[f"{tag}:temperature"]))(x1, x2)
    assert tuple(val.shape) == (x1.shape[0],)
    np.testing.assert_allclose(val.detach().cpu().numpy(), g[f"{tag}:value"], rtol=2e-5, atol=2e-6)
    (val * torch.from_numpy(g[f"{tag}:r"]).to(cuda)).sum().backward()
    _close(x1.grad, g[f"{tag}:grad1"], 2e-4, "grad1")
    _close(x2.grad, g[f"{tag}:grad2"], 2e-4, "grad2")


@pytest.mark.parametrize("shape", [(2, 32, 32, 32, 32), (5, 20, 9, 11, 13), (8, 64, 8, 8, 8),
                                   (2, 260, 4, 4, 4)])
def test_loco_loss_matches_cpu_oracle(cuda, shape):
    g = torch.Generator().manual_seed(shape[1])
    a = torch.randn(shape, generator=g)
    b = torch.randn(shape, generator=g) + 0.5 * a
    r = torch.rand(shape[0], generator=g) + 0.5
    ac, bc = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = local_contrastive_loss(ac, bc, 0.1)
    (ref * r).sum().backward()
    ah, bh = a.to(cuda).requires_grad_(True), b.to(cuda).requires_grad_(True)
    val = HF.loco_loss(ah, bh, 0.1)
    (val * r.to(cuda)).sum().backward()
    np.testing.assert_allclose(val.detach().cpu().numpy(), ref.detach().numpy(), rtol=2e-5, atol=2e-6)
    _close(ah.grad, ac.grad.numpy(), 2e-4, "grad1")
    _close(bh.grad, bc.grad.numpy(), 2e-4, "grad2")
    # deterministic (fixed-order fold of the block partials)
    assert torch.equal(val, HF.loco_loss(ah, bh, 0.1))
    # only the student's gradient when the teacher is detached
    ah.grad = None
    HF.loco_loss(ah, bh.detach(), 0.1).sum().backward()
    assert ah.grad is not None


def test_loco_loss_rejects_what_it_cannot_run(cuda):
    x = torch.randn(9, 8, 4, 4, 4, device=cuda)
    with pytest.raises(Exception, match="batch must be"):
        HF.loco_loss(x, x)
    y = torch.randn(2, 6, 4, 4, 4, device=cuda)
    with pytest.raises(Exception, match="C % 4"):
        HF.loco_loss(y, y)
    with pytest.raises(Exception, match="differ"):
        HF.loco_loss(torch.randn(2, 8, 4, 4, 4, device=cuda), torch.randn(2, 8, 4, 4, 5, device=cuda))


def _net(name, cuda, cls=UNetSemiSL, **extra):
    kw = dict(SEMISL_CASES[name])
    kw["activation_fn"] = activation_factory[kw["activation_fn"]]
    net = cls(**extra, **kw)
    sd = {k: torch.from_numpy(tensor_for(k, v.shape)) if v.numel() > 0 else v
          for k, v in net.state_dict().items() if "shadow" not in k}
    net.load_state_dict(sd, strict=False)
    return net.to(cuda).eval()


@pytest.mark.parametrize("name", list(SEMISL_CASES))
def test_unet_semisl_features_and_step_match_reference(cuda, name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    loss_fn = CompoundLoss([(binary_generalized_dice_loss, dict(smooth=1e-5, eps=1e-6)),
                            (binary_focal_loss, dict(gamma=1.0, eps=1e-6))])
    net = _net(name, cuda, UNetContrastiveSemiSL, loss_fn=loss_fn, label_key="label",
               loss_fn_semi_sl=LocalContrastiveLoss(seed=42))
    assert list(g["param_keys"]) == [k for k, _ in net.named_parameters()]
    x, x1, x2, y = (torch.from_numpy(g[k]).to(cuda) for k in ("x", "x1", "x2", "y"))
    f1 = net.forward_features(X=x1)
    f2 = net.forward_features_ema_stop_grad(X=x2, apply_linear_transformation=True)
    assert not f2.requires_grad
    _close(f1, g["features_1"], 1e-4, "features_1")
    _close(f2, g["features_2"], 1e-4, "features_2")
    _close(net.loss_fn_semi_sl(f1, f2), g["loco"], 1e-4, "loco")
    # forward(return_features=True) returns the same feature map (unet.py:171-174)
    _, feats, _ = net(x1, return_features=True)
    assert torch.equal(feats, f1)
    batch = {"supervised": {"image": x, "label": y},
             "self_supervised": {"semi_sl_image_1": x1, "semi_sl_image_2": x2}}
    net.zero_grad()
    total = net.training_step(batch, 0)
    _close(total, g["total"], 1e-4, "total")
    total.backward()
    for k, p in net.named_parameters():
        key = "grad:" + k
        if key not in g.files:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        ref = g[key]
        scale = max(np.abs(ref).max(), 1e-3 * max(np.abs(g["grad:" + kk]).max()
                                                  for kk in g["param_keys"] if "grad:" + kk in g.files))
        assert np.abs(p.grad.cpu().numpy() - ref).max() <= 2e-3 * scale, k
    # the self-supervised branch alone (label_key None, pl.py:395, 418-433)
    net.label_key = None
    only = net.training_step(batch, 0)
    _close(only, g["ssl"], 1e-4, "ssl only")


def test_unet_contrastive_semisl_with_ema_teacher_trains(cuda):
    """EMA teacher (network_factories.py:602-620): the shadow produces the view-2 features, is
    updated after every step by the fused EMA kernel and never receives gradients."""
    loss_fn = CompoundLoss([(binary_generalized_dice_loss, dict(smooth=1e-5, eps=1e-6)),
                            (binary_focal_loss, dict(gamma=1.0, eps=1e-6))])
    ema = ExponentialMovingAverage(decay=0.9, final_decay=1.0, n_steps=10)
    net = _net("unet3d_semisl", cuda, UNetContrastiveSemiSL, loss_fn=loss_fn, label_key="label",
               ema=ema, learning_rate=1e-2)
    net.train()
    assert net.ema.shadow is not None   # (train() on the parent also flips the shadow, as in the reference)
    g = torch.Generator().manual_seed(3)
    mk = lambda *s: torch.rand(*s, generator=g).to(cuda)  # noqa: E731
    batch = {"supervised": {"image": mk(2, 2, 16, 16, 16),
                            "label": (mk(2, 1, 16, 16, 16) > 0.8).float()},
             "self_supervised": {"semi_sl_image_1": mk(2, 2, 16, 16, 16),
                                 "semi_sl_image_2": mk(2, 2, 16, 16, 16)}}
    opt = net.configure_optimizers()["optimizer"]
    runner = StepRunner(net, opt, None)
    student0 = {k: p.detach().clone() for k, p in net.named_parameters() if "shadow" not in k}
    shadow0 = {k: p.detach().clone() for k, p in net.ema.shadow.named_parameters()}
    losses = [float(runner.train_step(batch).detach()) for _ in range(3)]
    assert all(np.isfinite(losses))
    moved = sum(not torch.equal(p.detach(), student0[k]) for k, p in net.named_parameters()
                if "shadow" not in k)
    assert moved > 0.5 * len(student0)
    # shadow = decay * shadow + (1 - decay) * student after every step: it moved, towards the student
    sh = dict(net.ema.shadow.named_parameters())
    k = "linear_transformation.weight"
    assert not torch.equal(sh[k], shadow0[k])
    st = dict(net.named_parameters())[k]
    assert float((sh[k] - st).abs().max()) < float((shadow0[k] - st).abs().max())
    assert all(p.grad is None for p in net.ema.shadow.parameters())
