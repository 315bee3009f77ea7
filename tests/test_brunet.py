"""Multi-branch U-Net (reference adell_mri/modules/segmentation/unet.py:846-1253)
against fixtures generated from the real reference (oracle/make_golden.py BRUNET_CASES), and the
concurrent squeeze-and-excite gate kernel against torch."""
import os

import numpy as np
import pytest
import torch

from adell_mri_amd.modules.activations import activation_factory
from adell_mri_amd.modules.segmentation.unet import BrUNet
from cases import BRUNET_CASES, grad_rel_err
from oracle.torch_ref.unet import compound_loss
from oracle.weights import fill_state_dict

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def build(name):
    kw = dict(BRUNET_CASES[name][0])
    kw["activation_fn"] = activation_factory[kw["activation_fn"]]
    net = BrUNet(**kw)
    net.load_state_dict(fill_state_dict(net.state_dict()))
    return net


@pytest.mark.parametrize("name", list(BRUNET_CASES))
def test_parameter_names_and_shapes_equal_reference(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    net = build(name)
    assert [k for k, _ in net.named_parameters()] == [str(k) for k in g["param_keys"]]
    assert [",".join(map(str, p.shape)) for _, p in net.named_parameters()] == \
        [str(s) for s in g["param_shapes"]]


def test_fix_input_fills_missing_items_and_zeroes_their_weights():
    a, b = torch.rand(2, 4, 4), torch.rand(2, 4, 4)
    X, w = BrUNet.fix_input([[a, None, b], [None, b, a]])
    assert [tuple(x.shape) for x in X] == [(3, 2, 4, 4)] * 2
    assert w[0].tolist() == [1.0, 0.0, 1.0] and w[1].tolist() == [0.0, 1.0, 1.0]
    assert torch.equal(X[0][1], torch.zeros(2, 4, 4)) and torch.equal(X[1][2], a)
    with pytest.raises(AssertionError):
        BrUNet.fix_input([[a, torch.rand(2, 4, 5)], [a, a]])


def test_cpu_input_is_refused():
    from adell_mri_amd._lib import AdellHipError
    with pytest.raises(AdellHipError):
        build("brunet2d_missing_inputs")([torch.rand(1, 2, 32, 40)] * 2)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(BRUNET_CASES))
def test_brunet_matches_reference(cuda, name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    net = build(name).to(cuda).eval()
    xs = [torch.from_numpy(g["x0"]).to(cuda), torch.from_numpy(g["x1"]).to(cuda)]
    ws = [torch.from_numpy(g["w0"]).to(cuda), torch.from_numpy(g["w1"]).to(cuda)] \
        if "w0" in g.files else None
    y = torch.from_numpy(g["y"]).to(cuda)
    bott = net(xs, ws, return_bottleneck=True)[2]
    ref = g["bottleneck"]
    assert np.abs(bott.detach().cpu().numpy() - ref).max() / np.abs(ref).max() < 1e-4
    logits, _ = net(xs, ws, return_logits=True)
    ref = g["logits"]
    assert np.abs(logits.detach().cpu().numpy() - ref).max() / np.abs(ref).max() < 1e-4
    prob, _ = net(xs, ws)
    loss = compound_loss(prob, y)
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-4)
    loss.backward()
    for k, p in net.named_parameters():
        if ("grad:" + k) not in g.files:
            continue
        assert p.grad is not None, k
        assert grad_rel_err(g, k, p.grad.cpu().numpy()) < 3e-3, k


@pytest.mark.gpu
@pytest.mark.parametrize("shape,use_inv,use_acc", [((2, 24, 5, 6, 7), True, True),
                                                   ((3, 8, 4, 9, 3), False, False),
                                                   ((1, 100, 3, 5, 6), True, False),
                                                   ((2, 64, 2, 8, 8), False, True),
                                                   ((2, 300, 1, 6, 5), True, True)])
def test_cse_apply_fwd_bwd_matches_torch(cuda, shape, use_inv, use_acc):
    from adell_mri_amd import functional as HF
    from adell_mri_amd import ops
    g = torch.Generator().manual_seed(shape[1])
    N, C = shape[:2]
    x = torch.randn(shape, generator=g, dtype=torch.float64).requires_grad_(True)
    s = torch.rand((N, 1, *shape[2:]), generator=g, dtype=torch.float64).requires_grad_(True)
    c = torch.rand((N, C), generator=g, dtype=torch.float64).requires_grad_(True)
    inv = (torch.rand(N, generator=g, dtype=torch.float64) + 0.5) if use_inv else None
    acc = torch.randn(shape, generator=g, dtype=torch.float64).requires_grad_(True) if use_acc else None
    y = x * (s + c[:, :, None, None, None])
    if use_inv:
        y = y * inv[:, None, None, None, None]
    if use_acc:
        y = y + acc
    dy = torch.randn(shape, generator=g, dtype=torch.float64)
    y.backward(dy)
    dev = lambda t: None if t is None else t.detach().float().to(cuda)  # noqa: E731
    xd = ops.ndhwc(dev(x)).requires_grad_(True)
    sd, cd = dev(s).requires_grad_(True), dev(c).requires_grad_(True)
    accd = None if acc is None else ops.ndhwc(dev(acc)).requires_grad_(True)
    yd = HF.cse_apply(xd, sd, cd, dev(inv), accd)
    assert float((yd.detach().cpu().double() - y.detach()).abs().max()) < 1e-5
    yd.backward(ops.ndhwc(dev(dy)))
    tol = lambda ref: 2e-5 * max(1.0, float(ref.abs().max()))  # noqa: E731
    assert float((xd.grad.cpu().double() - x.grad).abs().max()) < tol(x.grad)
    assert float((sd.grad.cpu().double() - s.grad).abs().max()) < tol(s.grad)
    assert float((cd.grad.cpu().double() - c.grad).abs().max()) < tol(c.grad)
    if use_acc:
        assert torch.equal(accd.grad.cpu(), dy.float())


@pytest.mark.gpu
def test_channel_mean_fwd_bwd_matches_torch(cuda):
    from adell_mri_amd import functional as HF
    from adell_mri_amd import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn((3, 20, 4, 5, 6), generator=g, dtype=torch.float64).requires_grad_(True)
    m = x.flatten(2).mean(-1)
    dm = torch.randn(m.shape, generator=g, dtype=torch.float64)
    m.backward(dm)
    xd = ops.ndhwc(x.detach().float().to(cuda)).requires_grad_(True)
    md = HF.channel_mean(xd)
    assert float((md.detach().cpu().double() - m.detach()).abs().max()) < 1e-6
    md.backward(dm.float().to(cuda))
    assert float((xd.grad.cpu().double() - x.grad).abs().max()) < 1e-7


@pytest.mark.gpu
def test_brunetpl_training_step_matches_oracle_sgd_step(cuda):
    """One BrUNetPL step (forward with branch weights, dice + focal loss, backward, fused
    SGD-Nesterov) against the CPU oracle stepped by torch.optim.SGD on the same weights."""
    from adell_mri_amd.modules.segmentation.losses import (CompoundLoss, binary_focal_loss,
                                                           binary_generalized_dice_loss)
    from adell_mri_amd.modules.segmentation.pl import BrUNetPL
    from adell_mri_amd.trainer import StepRunner
    from cases import oracle_cfg
    from oracle.torch_ref.brunet import BrUNetOracle
    name = "brunet3d_two_branch"
    g = np.load(os.path.join(GOLD, name + ".npz"))
    kw = dict(BRUNET_CASES[name][0])
    loss_fn = CompoundLoss([(binary_generalized_dice_loss, {"smooth": 1e-5, "eps": 1e-6}),
                            (binary_focal_loss, {"gamma": 1.0, "eps": 1e-6})])
    net = BrUNetPL(image_keys=["t2", "adc"], label_key="mask", loss_fn=loss_fn, learning_rate=5e-4,
                   weight_decay=5e-3, **dict(kw, activation_fn=activation_factory["swish"]))
    net.load_state_dict(fill_state_dict(net.state_dict()))
    w = [torch.tensor([1.0, 0.5]), torch.tensor([0.25, 1.0])]
    batch = {"t2": torch.from_numpy(g["x0"]), "adc": torch.from_numpy(g["x1"]),
             "t2_weight": w[0], "adc_weight": w[1], "mask": torch.from_numpy(g["y"])}
    # CPU side: oracle on the same state dict, torch SGD-Nesterov (pl.py:563-569)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items() if v.dtype == torch.float32}
    oracle = BrUNetOracle(sd, oracle_cfg(dict(kw, n_classes=2))).requires_grad_(True)   # own copies
    params = [oracle.sd[k] for k, _ in net.named_parameters()]
    opt_ref = torch.optim.SGD(params, lr=5e-4, momentum=0.99, weight_decay=5e-3, nesterov=True)
    prob = torch.sigmoid(oracle.forward([batch["t2"], batch["adc"]], w))
    loss_ref = compound_loss(prob, batch["mask"])
    loss_ref.backward()
    opt_ref.step()
    # MI355X side
    net = net.to(cuda).eval()   # eval(): dropout_param is 0, instance norm has no running stats
    opt = net.configure_optimizers()["optimizer"]
    loss = StepRunner(net, opt).train_step({k: v.to(cuda) for k, v in batch.items()})
    np.testing.assert_allclose(loss.item(), loss_ref.item(), rtol=1e-4)
    for (k, p), r in zip(net.named_parameters(), params):
        step_ref = r.detach() - sd[k]                 # what the reference step changed
        step = p.detach().cpu() - sd[k]
        scale = max(step_ref.abs().max().item(), 1e-9)
        assert (step - step_ref).abs().max().item() <= 5e-3 * scale + 1e-8, k
