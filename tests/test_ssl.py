"""ConvNeXt backbone + VICReg self-supervised step (SURVEY.md 8 rows a13 / a15;
BASELINE config 4 in miniature) against fixtures generated from the real reference
(oracle/make_golden.py gen_ssl)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from adell_mri_amd import functional as HF
from cases import grad_rel_err
from oracle.torch_ref.convnext import ConvNeXtOracle, vicreg_loss
from oracle.weights import fill_state_dict, tensor_for

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SSL_CASE = dict(
    backbone_args=dict(spatial_dim=3, in_channels=1, structure=[[8, 16, 3, 2], [16, 32, 7, 2]],
                       maxpool_structure=[2, 2]),
    projection_head_args=dict(in_channels=16, structure=[32, 24]),
    prediction_head_args=dict(in_channels=24, structure=[32, 24]))
SSL_GAIN = 3.0
SSL_OPT = dict(learning_rate=1e-3, weight_decay=5e-3, optimizer_eps=1e-8)
ORACLE_CFG = dict(structure=SSL_CASE["backbone_args"]["structure"], maxpool_structure=[2, 2],
                  projection_structure=[32, 24], prediction_structure=[32, 24])


def gold():
    return np.load(os.path.join(GOLD, "ssl_convnext_small.npz"), allow_pickle=False)


def rel(a, r):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else a
    return float(np.abs(a - r).max() / (np.abs(r).max() + 1e-30))


def build_pl(ema=None):
    from adell_mri_amd.modules.layers.adn_fn import get_adn_fn
    from adell_mri_amd.modules.self_supervised.pl import SelfSLConvNeXtPL

    adn1 = get_adn_fn(1, "layer", "gelu", 0.0)
    kw = {k: dict(v) for k, v in SSL_CASE.items()}
    kw["projection_head_args"]["adn_fn"] = adn1
    kw["prediction_head_args"]["adn_fn"] = adn1
    net = SelfSLConvNeXtPL(aug_image_key_1="a", aug_image_key_2="b", ssl_method="vicreg",
                           stop_gradient=False, n_epochs=10, batch_size=4, ema=None,
                           **SSL_OPT, **kw)
    net.load_state_dict(fill_state_dict(net.state_dict(), gain=SSL_GAIN))
    if ema is not None:  # as the constructor does (pl.py:887-889), after the weights are set
        net.ema = ema
        ema.update(net)
    return net


# ---- CPU: the oracle is pinned to the reference; the product mirrors its interface -------
def test_convnext_oracle_matches_reference():
    g = gold()
    sd = {str(k): torch.from_numpy(tensor_for(str(k), g["grad:" + str(k)].shape))
          for k in g["param_keys"]}
    sd = {k: (v * SSL_GAIN if (v.dim() > 1 and v.shape[0] > 1) else v) for k, v in sd.items()}
    net = ConvNeXtOracle(sd, ORACLE_CFG).requires_grad_(True)
    x1, x2 = torch.from_numpy(g["x1"]), torch.from_numpy(g["x2"])
    assert rel(net.forward(x1, "representation"), g["representation"]) < 1e-5
    y1, y2 = net.forward(x1, "prediction"), net.forward(x2, "projection")
    assert rel(y1, g["y1"]) < 1e-5 and rel(y2, g["y2"]) < 1e-5
    losses = vicreg_loss(y1, y2)
    np.testing.assert_allclose(torch.stack(losses).detach().numpy(), g["losses"], rtol=2e-4,
                               atol=1e-9)
    sum(losses).backward()
    for k, p in net.sd.items():
        assert grad_rel_err(g, k, p.grad.numpy()) < 2e-3, k


def test_vicreg_oracle_matches_reference():
    g = gold()
    x1 = torch.from_numpy(g["vic_x1"]).requires_grad_(True)
    x2 = torch.from_numpy(g["vic_x2"]).requires_grad_(True)
    terms = vicreg_loss(x1, x2)
    np.testing.assert_allclose(torch.stack(terms).detach().numpy(), g["vic_terms"], rtol=1e-5)
    sum(terms).backward()
    assert rel(x1.grad, g["vic_dx1"]) < 1e-5 and rel(x2.grad, g["vic_dx2"]) < 1e-5


def test_ssl_module_state_dict_keys_equal_reference():
    g = gold()
    net = build_pl()
    assert [k for k, _ in net.named_parameters()] == [str(k) for k in g["param_keys"]]
    for k, p in net.named_parameters():
        assert tuple(p.shape) == g["grad:" + k].shape, k


def test_ssl_module_fails_loudly_without_gpu():
    net = build_pl()
    with pytest.raises(Exception):
        net(torch.zeros((2, 1, 32, 32, 32)), ret="projection")


def test_unsupported_ssl_methods_raise():
    from adell_mri_amd.modules.self_supervised.pl import SelfSLConvNeXtPL

    with pytest.raises(NotImplementedError):
        SelfSLConvNeXtPL(ssl_method="vicregl", **SSL_CASE)


def test_loss_selection_follows_the_reference():
    """self_supervised/pl.py:202-212."""
    from adell_mri_amd.modules.self_supervised.losses import (NTXentLoss, VICRegLoss, byol_loss,
                                                              simsiam_loss)
    from adell_mri_amd.modules.self_supervised.pl import SelfSLConvNeXtPL
    from adell_mri_amd.utils import ExponentialMovingAverage

    assert SelfSLConvNeXtPL(ssl_method="simsiam", **SSL_CASE).loss is simsiam_loss
    assert SelfSLConvNeXtPL(ssl_method="byol", ema=ExponentialMovingAverage(0.99),
                            **SSL_CASE).loss is byol_loss
    simclr = SelfSLConvNeXtPL(ssl_method="simclr", temperature=0.5, **SSL_CASE).loss
    assert isinstance(simclr, NTXentLoss) and simclr.temperature == 0.5 and simclr.apply_relu
    assert isinstance(SelfSLConvNeXtPL(ssl_method="vicreg", **SSL_CASE).loss, VICRegLoss)


# ---- GPU: HIP kernels against stock torch / the reference fixtures ---------------------
@pytest.mark.gpu
@pytest.mark.parametrize("N,C,size,k", [(2, 8, (6, 6, 6), 3), (1, 96, (16, 16, 16), 7),
                                        (2, 24, (4, 5, 3), 7), (1, 130, (2, 2, 2), 3),
                                        (1, 8, (7, 6, 9), (3, 1, 5)),
                                        (1, 20, (5, 9, 21), 7), (1, 16, (6, 6, 37), 3),
                                        (2, 8, (9, 10, 11), 5), (1, 33, (3, 17, 16), 5),
                                        # 7^3 with rows of 9 ... 16 voxels: the z-marching kernel (ragged
                                        # channels / rows, several z segments)
                                        (3, 24, (10, 11, 13), 7), (1, 16, (40, 9, 16), 7)])
def test_depthwise_conv3d_fwd_bwd(cuda, N, C, size, k):
    from adell_mri_amd import functional as HF
    from adell_mri_amd import ops

    ks = (k, k, k) if isinstance(k, int) else k
    g = torch.Generator().manual_seed(3)
    x = torch.randn((N, C, *size), generator=g, dtype=torch.float64).requires_grad_(True)
    w = (torch.randn((C, 1, *ks), generator=g, dtype=torch.float64) / np.sqrt(np.prod(ks))
         ).requires_grad_(True)
    b = torch.randn((C,), generator=g, dtype=torch.float64).requires_grad_(True)
    y = F.conv3d(x, w, b, padding=[q // 2 for q in ks], groups=C)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(dy)
    xd = ops.ndhwc(x.detach().float().to(cuda)).requires_grad_(True)
    wd = w.detach().float().to(cuda).requires_grad_(True)
    bd = b.detach().float().to(cuda).requires_grad_(True)
    yd = HF.dwconv3d(xd, wd, bd)
    yd.backward(ops.ndhwc(dy.float().to(cuda)))
    assert rel(yd, y.detach().numpy()) < 2e-6
    assert rel(xd.grad, x.grad.numpy()) < 2e-6
    assert rel(wd.grad, w.grad.numpy()) < 5e-6
    assert rel(bd.grad, b.grad.numpy()) < 5e-6


@pytest.mark.gpu
def test_vicreg_loss_terms_and_grads_match_reference(cuda):
    from adell_mri_amd.modules.self_supervised.losses import VICRegLoss

    g = gold()
    x1 = torch.from_numpy(g["vic_x1"]).to(cuda).requires_grad_(True)
    x2 = torch.from_numpy(g["vic_x2"]).to(cuda).requires_grad_(True)
    terms = VICRegLoss()(x1, x2)
    np.testing.assert_allclose(torch.stack(terms).detach().cpu().numpy(), g["vic_terms"],
                               rtol=2e-5)
    sum(terms).backward()
    assert rel(x1.grad, g["vic_dx1"]) < 2e-5 and rel(x2.grad, g["vic_dx2"]) < 2e-5


@pytest.mark.gpu
# (B <= 64 and D >= 256: the forward on one block per 128 feature columns + a fold; else one block)
@pytest.mark.parametrize("B,D", [(2, 7), (16, 1024), (5, 300), (32, 2048), (64, 520), (65, 512)])
def test_vicreg_loss_matches_oracle_at_other_sizes(cuda, B, D):
    from adell_mri_amd.modules.self_supervised.losses import VICRegLoss

    gen = torch.Generator().manual_seed(B * D)
    a = torch.randn((B, D), generator=gen, dtype=torch.float64).requires_grad_(True)
    b = (0.3 * a.detach() + torch.randn((B, D), generator=gen, dtype=torch.float64)
         ).requires_grad_(True)
    ref = vicreg_loss(a, b)
    sum(ref).backward()
    x1 = a.detach().float().to(cuda).requires_grad_(True)
    x2 = b.detach().float().to(cuda).requires_grad_(True)
    terms = VICRegLoss()(x1, x2)
    np.testing.assert_allclose(torch.stack(terms).detach().cpu().numpy(),
                               torch.stack(ref).detach().numpy(), rtol=5e-5)
    sum(terms).backward()
    assert rel(x1.grad, a.grad.numpy()) < 5e-5 and rel(x2.grad, b.grad.numpy()) < 5e-5


@pytest.mark.gpu
@pytest.mark.parametrize("tag,k,oc", [("k3", 3, 8), ("k7", 7, 12)])
def test_convnext_block_matches_reference(cuda, tag, k, oc):
    from adell_mri_amd.modules.layers.res_blocks import ConvNeXtBlock3d

    g = gold()
    blk = ConvNeXtBlock3d(8, k, 16, oc)
    blk.load_state_dict(fill_state_dict(blk.state_dict()))
    blk = blk.to(cuda)
    x = torch.from_numpy(g["blk_x"]).to(cuda).requires_grad_(True)
    y = blk(x)
    assert rel(y, g[f"blk_{tag}_y"]) < 1e-5
    (y * torch.from_numpy(g[f"blk_{tag}_r"]).to(cuda)).sum().backward()
    assert rel(x.grad, g[f"blk_{tag}_dx"]) < 1e-4
    for n, p in blk.named_parameters():
        ref = g[f"blk_{tag}_grad:{n}"]
        assert rel(p.grad, ref) < 1e-4, n


@pytest.mark.gpu
def test_vicreg_training_step_matches_reference(cuda):
    """forward of both views, VICReg terms, every parameter gradient, one AdamW step and the
    EMA shadow update (self_supervised/pl.py:904-972, :231-267; utils/utils.py:447-493)."""
    from adell_mri_amd.trainer import StepRunner
    from adell_mri_amd.utils import ExponentialMovingAverage

    g = gold()
    net = build_pl().to(cuda).train()
    net.ema = ExponentialMovingAverage(0.99)
    net.ema.update(net)
    x1, x2 = torch.from_numpy(g["x1"]).to(cuda), torch.from_numpy(g["x2"]).to(cuda)
    assert rel(net(x1, ret="representation"), g["representation"]) < 1e-4
    assert rel(net(x1, ret="prediction"), g["y1"]) < 1e-4
    assert rel(net(x2, ret="projection"), g["y2"]) < 1e-4

    # the fixture's second view goes through the online network (no EMA forward): keep the
    # shadow for the update only
    ema, net.ema = net.ema, None
    runner = StepRunner(net)
    net.ema = None
    runner.optimizer.zero_grad()
    loss = net.training_step({"a": x1, "b": x2}, 0)
    np.testing.assert_allclose(torch.stack(list(net.last_losses)).detach().cpu().numpy(),
                               g["losses"], rtol=5e-4, atol=1e-8)
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-4)
    loss.backward()
    for k, p in net.named_parameters():
        assert grad_rel_err(g, k, p.grad.cpu().numpy()) < 5e-3, k
    runner.optimizer.step()
    ema.update(net)
    for k, p in net.named_parameters():
        np.testing.assert_allclose(p.detach().cpu().numpy(), g["step1:" + k], rtol=2e-4,
                                   atol=2e-6, err_msg=k)
    assert ema._plan is not None, "EMA should take the single-launch flat path"
    for k, p in ema.shadow.named_parameters():
        np.testing.assert_allclose(p.detach().cpu().numpy(), g["ema1:" + k], rtol=1e-5,
                                   atol=1e-7, err_msg=k)


@pytest.mark.gpu
def test_vicreg_one_pass_over_both_views_equals_two_passes(cuda):
    """Without batch-coupled layers the two views share one pass of 2B items: the same loss
    terms and gradients as the two separate passes (which the EMA / stop-gradient / batch-norm
    configurations keep)."""
    g = gold()
    batch = {"a": torch.from_numpy(g["x1"]).to(cuda), "b": torch.from_numpy(g["x2"]).to(cuda)}
    out = {}
    for mode in ("one", "two"):
        net = build_pl().to(cuda).train()
        assert net._views_share_a_pass("prediction", "projection") is True
        if mode == "two":
            net._batch_coupled = True      # what a BatchNorm layer anywhere in the module sets
            assert net._views_share_a_pass("prediction", "projection") is False
        loss = net.training_step(batch, 0)
        loss.backward()
        out[mode] = (torch.stack(list(net.last_losses)).detach(),
                     {k: p.grad.detach().clone() for k, p in net.named_parameters()})
    np.testing.assert_allclose(out["one"][0].cpu().numpy(), out["two"][0].cpu().numpy(), rtol=1e-5)
    gmax = max(float(v.abs().max()) for v in out["two"][1].values())
    for k, v in out["two"][1].items():
        assert float((out["one"][1][k] - v).abs().max()) <= 1e-4 * max(float(v.abs().max()),
                                                                        1e-3 * gmax), k
    # a target branch of its own (EMA or stop-gradient) keeps the two passes
    net = build_pl().to(cuda)
    net.stop_gradient = True
    assert net._views_share_a_pass("prediction", "projection") is False


@pytest.mark.gpu
def test_ema_forward_and_stop_gradient_step_runs(cuda):
    """BYOL-style wiring: second view through the EMA shadow under no_grad; the loss
    decreases over a few steps and the shadow trails the online weights."""
    from adell_mri_amd.trainer import StepRunner
    from adell_mri_amd.utils import ExponentialMovingAverage

    g = gold()
    net = build_pl().to(cuda).train()
    net.stop_gradient = True
    net.ema = ExponentialMovingAverage(0.9)
    net.ema.update(net)
    runner = StepRunner(net)
    batch = {"a": torch.from_numpy(g["x1"]).to(cuda), "b": torch.from_numpy(g["x2"]).to(cuda)}
    w0 = net.backbone.input_layer[0].weight.detach().clone()
    losses = [runner.train_step(batch).item() for _ in range(4)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    w1 = net.backbone.input_layer[0].weight.detach()
    ws = net.ema.shadow.backbone.input_layer[0].weight.detach()
    assert not torch.equal(w0, w1)
    # shadow lies strictly between the initial and the current weights
    d_total = (w1 - w0).abs().max().item()
    assert 0 < (ws - w0).abs().max().item() < d_total
    for _, p in net.ema.shadow.named_parameters():
        assert p.grad is None and not p.requires_grad


@pytest.mark.gpu
def test_selfsl_unet_vicreg_step(cuda):
    """SelfSLUNetPL (self_supervised/pl.py:538-756): the U-Net encoder alone as the SSL backbone;
    VICReg on the spatial means of the two bottleneck maps. The loss terms are checked against
    the CPU restatement of VICRegLoss evaluated on the HIP bottlenecks, the gradient against a
    finite difference along a random direction, and one fused AdamW step must move the encoder."""
    from adell_mri_amd.modules.activations import activation_factory
    from adell_mri_amd.modules.self_supervised.pl import SelfSLUNetPL
    from adell_mri_amd.trainer import StepRunner
    from oracle.torch_ref.convnext import vicreg_loss

    torch.manual_seed(0)
    net = SelfSLUNetPL(aug_image_key_1="a", aug_image_key_2="b", ssl_method="vicreg",
                       stop_gradient=False, learning_rate=1e-3, weight_decay=1e-3,
                       spatial_dimensions=3, depth=[8, 16, 32], kernel_sizes=[3, 3, 3],
                       strides=[2, 2, 2], norm_type="instance", padding=1, in_channels=1,
                       activation_fn=activation_factory["swish"], dropout_param=0.0).to(cuda).train()
    assert net.encoder_only is True and not hasattr(net, "decoding_operations")
    g = torch.Generator().manual_seed(3)
    zz = torch.arange(16.0)[None, None, :, None, None]
    x1 = torch.stack([torch.sin((b + 1) * 0.4 * zz[0]) + 0.3 * torch.rand((1, 16, 16, 16), generator=g)
                      for b in range(6)])
    x2 = x1 + 0.2 * torch.randn(x1.shape, generator=g)
    batch = {"a": x1.to(cuda), "b": x2.to(cuda)}
    loss = net.training_step(batch, 0)
    y1, y2 = net(batch["a"]), net(batch["b"])
    assert y1.shape == (6, 32, 4, 4, 4)
    m1, m2 = y1.flatten(2).mean(-1).detach().cpu(), y2.flatten(2).mean(-1).detach().cpu()
    want = vicreg_loss(m1, m2)       # (inv, var, cov) weighted 25 / 25 / 0.1
    got = [float(t.detach()) for t in net.last_losses]
    np.testing.assert_allclose(got, [float(t) for t in want], rtol=2e-4, atol=1e-6)
    np.testing.assert_allclose(float(loss.detach()), sum(got), rtol=1e-6)
    # ... and INDEPENDENTLY of the HIP path: the stock-torch CPU oracle of the U-Net encoder
    # (oracle/torch_ref/unet.py, pinned to the reference fixtures) on the same weights and inputs
    # gives the bottlenecks, the three terms and every parameter gradient
    from oracle.torch_ref.unet import UNetOracle
    ref = UNetOracle({k: v.detach().cpu() for k, v in net.state_dict().items()},
                     dict(depth=[8, 16, 32], kernel_sizes=[3, 3, 3], strides=[2, 2, 2], padding=1,
                          norm_type="instance", activation="swish")).requires_grad_(True)
    b1 = ref.encode("encoding_operations", x1)[1]
    b2 = ref.encode("encoding_operations", x2)[1]
    scale = float(b1.detach().abs().max())
    assert float((y1.detach().cpu() - b1.detach()).abs().max()) < 1e-4 * scale
    terms = vicreg_loss(b1.flatten(2).mean(-1), b2.flatten(2).mean(-1))
    np.testing.assert_allclose(got, [float(t) for t in terms], rtol=5e-4, atol=1e-6)
    sum(terms).backward()
    net.zero_grad()
    net.training_step(batch, 0).backward()
    top = max(float(v.grad.abs().max()) for v in ref.sd.values() if v.grad is not None)
    for k, p in net.named_parameters():
        rg = ref.sd[k].grad
        if rg is None:
            continue
        err = float((p.grad.detach().cpu() - rg).abs().max())
        assert err <= 3e-3 * max(float(rg.abs().max()), 1e-2 * top), (k, err)
    net.zero_grad()
    before = {k: p.detach().clone() for k, p in net.named_parameters()}
    StepRunner(net).train_step(batch)
    moved = sum(not torch.equal(before[k], p.detach()) for k, p in net.named_parameters())
    assert moved > 0.8 * len(before)


# ---- 2-D ConvNeXt (the reference's sample_configs/ssl-2d-convnext.yaml) --------------------------
SSL2D_CASE = dict(
    backbone_args=dict(spatial_dim=2, in_channels=1, structure=[[8, 16, 7, 2], [16, 32, 3, 2]],
                       maxpool_structure=[[2, 2], [2, 2]], first_layer_stride=4),
    projection_head_args=dict(in_channels=16, structure=[32, 24]),
    prediction_head_args=dict(in_channels=24, structure=[32, 24]))


def gold2d():
    return np.load(os.path.join(GOLD, "ssl_convnext2d_small.npz"), allow_pickle=False)


def build_pl2d():
    from adell_mri_amd.modules.layers.adn_fn import get_adn_fn
    from adell_mri_amd.modules.self_supervised.pl import SelfSLConvNeXtPL

    adn1 = get_adn_fn(1, "layer", "gelu", 0.0)
    kw = {k: dict(v) for k, v in SSL2D_CASE.items()}
    kw["projection_head_args"]["adn_fn"] = adn1
    kw["prediction_head_args"]["adn_fn"] = adn1
    net = SelfSLConvNeXtPL(aug_image_key_1="a", aug_image_key_2="b", ssl_method="vicreg",
                           stop_gradient=False, n_epochs=10, batch_size=4, ema=None,
                           **SSL_OPT, **kw)
    net.load_state_dict(fill_state_dict(net.state_dict(), gain=SSL_GAIN))
    return net


def test_convnext2d_state_dict_shapes_equal_reference():
    g = gold2d()
    net = build_pl2d()
    mine = {k: ",".join(map(str, p.shape)) for k, p in net.named_parameters()}
    assert list(mine) == list(g["param_keys"])
    assert list(mine.values()) == list(g["param_shapes"])


@pytest.mark.gpu
@pytest.mark.parametrize("tag,k,oc", [("k3", 3, 8), ("k7", 7, 12)])
def test_convnext_block2d_matches_reference(cuda, tag, k, oc):
    from adell_mri_amd.modules.layers.res_blocks import ConvNeXtBlock2d

    g = gold2d()
    blk = ConvNeXtBlock2d(8, k, 16, oc)
    blk.load_state_dict(fill_state_dict(blk.state_dict()))
    blk = blk.to(cuda)
    x = torch.from_numpy(g["blk_x"]).to(cuda).requires_grad_(True)
    y = blk(x)
    assert y.dim() == 4 and rel(y, g[f"blk_{tag}_y"]) < 1e-5
    (y * torch.from_numpy(g[f"blk_{tag}_r"]).to(cuda)).sum().backward()
    assert rel(x.grad, g[f"blk_{tag}_dx"]) < 1e-4
    for n, p in blk.named_parameters():
        assert rel(p.grad, g[f"blk_{tag}_grad:{n}"]) < 1e-4, n


@pytest.mark.gpu
def test_convnext2d_vicreg_step_matches_reference(cuda):
    """The three heads, the VICReg terms and every parameter gradient of the 2-D network, then one
    fused AdamW step through the PL wrapper."""
    from adell_mri_amd.trainer import StepRunner

    g = gold2d()
    net = build_pl2d().to(cuda).train()
    x1, x2 = torch.from_numpy(g["x1"]).to(cuda), torch.from_numpy(g["x2"]).to(cuda)
    assert rel(net(x1, ret="representation"), g["representation"]) < 1e-4
    assert rel(net(x1, ret="prediction"), g["y1"]) < 1e-4
    assert rel(net(x2, ret="projection"), g["y2"]) < 1e-4
    runner = StepRunner(net)
    runner.optimizer.zero_grad()
    loss = net.training_step({"a": x1, "b": x2}, 0)
    np.testing.assert_allclose(torch.stack(list(net.last_losses)).detach().cpu().numpy(),
                               g["losses"], rtol=5e-4, atol=1e-6)
    loss.backward()
    for k, p in net.named_parameters():
        assert grad_rel_err(g, k, p.grad.cpu().numpy()) < 5e-3, k
    before = net.backbone.input_layer[0].weight.detach().clone()
    runner.optimizer.step()
    assert not torch.equal(before, net.backbone.input_layer[0].weight.detach())


# ---- SimSiam / BYOL / NT-Xent losses ------------------------------------------------------------
def test_pair_loss_restatement_matches_reference_fixture():
    from oracle.torch_ref.ssl_losses import pair_loss

    g = np.load(os.path.join(GOLD, "ssl_pair_losses.npz"))
    for tag in [k[:-6] for k in g.files if k.endswith(":value")]:
        kind, temp, relu = str(g[f"{tag}:kind"]), float(g[f"{tag}:temperature"]), bool(g[f"{tag}:relu"])
        x1 = torch.from_numpy(g[f"{tag}:x1"]).requires_grad_(True)
        x2 = torch.from_numpy(g[f"{tag}:x2"]).requires_grad_(True)
        val = pair_loss(x1, x2, kind, temp, relu)
        val.backward()
        np.testing.assert_allclose(val.item(), g[f"{tag}:value"], rtol=2e-5, atol=1e-6)
        assert rel(x1.grad, g[f"{tag}:grad1"]) < 1e-4 and rel(x2.grad, g[f"{tag}:grad2"]) < 1e-4, tag


@pytest.mark.gpu
def test_pair_losses_match_reference_fixture(cuda):
    """simsiam_loss / byol_loss / NTXentLoss against values and gradients from the reference's
    own functions (oracle/make_golden.py gen_pair_losses)."""
    from adell_mri_amd.modules.self_supervised.losses import NTXentLoss, byol_loss, simsiam_loss

    g = np.load(os.path.join(GOLD, "ssl_pair_losses.npz"))
    tags = [k[:-6] for k in g.files if k.endswith(":value")]
    assert len(tags) >= 6
    for tag in tags:
        kind, temp, relu = str(g[f"{tag}:kind"]), float(g[f"{tag}:temperature"]), bool(g[f"{tag}:relu"])
        x1 = torch.from_numpy(g[f"{tag}:x1"]).to(cuda).requires_grad_(True)
        x2 = torch.from_numpy(g[f"{tag}:x2"]).to(cuda).requires_grad_(True)
        fn = {"simsiam": simsiam_loss, "byol": byol_loss,
              "ntxent": NTXentLoss(temperature=temp, apply_relu=relu)}[kind]
        val = fn(x1, x2)
        assert val.dim() == 0
        np.testing.assert_allclose(val.item(), g[f"{tag}:value"], rtol=3e-5, atol=2e-6)
        (val * 1.7).backward()
        assert rel(x1.grad, 1.7 * g[f"{tag}:grad1"]) < 2e-4, tag
        assert rel(x2.grad, 1.7 * g[f"{tag}:grad2"]) < 2e-4, tag


@pytest.mark.gpu
@pytest.mark.parametrize("kind,B,D", [("simsiam", 16, 1024), ("byol", 3, 7), ("ntxent", 32, 2048),
                                      ("ntxent", 128, 65), ("simsiam", 1, 5000)])
def test_pair_losses_match_oracle_at_other_sizes(cuda, kind, B, D):
    from oracle.torch_ref.ssl_losses import pair_loss

    g = torch.Generator().manual_seed(B + D)
    a = torch.randn(B, D, generator=g)
    b = 0.6 * a + 0.8 * torch.randn(B, D, generator=g)
    ac, bc = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = pair_loss(ac, bc, kind, 0.3, True)
    ref.backward()
    ah, bh = a.to(cuda).requires_grad_(True), b.to(cuda).requires_grad_(True)
    val = HF.pair_loss(ah, bh, kind, 0.3, True)
    val.backward()
    np.testing.assert_allclose(val.item(), ref.item(), rtol=3e-5, atol=2e-6)
    assert rel(ah.grad, ac.grad.numpy()) < 2e-4 and rel(bh.grad, bc.grad.numpy()) < 2e-4
    # stop-gradient: only the first argument gets a gradient, the same one
    a2 = a.to(cuda).requires_grad_(True)
    HF.pair_loss(a2, b.to(cuda), kind, 0.3, True).backward()
    assert torch.equal(a2.grad, ah.grad)


@pytest.mark.gpu
def test_byol_step_with_ema_target_trains(cuda):
    """ssl_method='byol': prediction of view 1 against the EMA shadow's projection of view 2 and
    the swapped pair (pl.py:456-500), cosine loss on the fused kernel."""
    from adell_mri_amd.modules.self_supervised.losses import byol_loss
    from adell_mri_amd.trainer import StepRunner
    from adell_mri_amd.utils import ExponentialMovingAverage

    g = gold()
    net = build_pl().to(cuda).train()
    net.ssl_method, net.stop_gradient = "byol", True
    net.ema = ExponentialMovingAverage(0.9)
    net.ema.update(net)
    del net.loss          # a VICRegLoss module was registered under this name
    net.init_loss()
    assert net.loss is byol_loss
    runner = StepRunner(net)
    batch = {"a": torch.from_numpy(g["x1"]).to(cuda), "b": torch.from_numpy(g["x2"]).to(cuda)}
    losses = [runner.train_step(batch).item() for _ in range(4)]
    assert all(np.isfinite(losses)) and 0.0 <= min(losses) and max(losses) <= 8.0
    assert losses[-1] < losses[0]


# ---- 2-D ResNet backbone (what vicreg / simclr / byol build from a 2-D backbone configuration) ---
@pytest.mark.gpu
def test_resnet2d_vicreg_step_matches_reference(cuda):
    from adell_mri_amd.modules.layers.adn_fn import get_adn_fn
    from adell_mri_amd.modules.self_supervised.pl import SelfSLResNetPL

    g = np.load(os.path.join(GOLD, "ssl_resnet2d_small.npz"), allow_pickle=False)
    adn, adn1 = get_adn_fn(2, "batch", "swish", 0.0), get_adn_fn(1, "layer", "gelu", 0.0)
    net = SelfSLResNetPL(
        aug_image_key_1="a", aug_image_key_2="b", ssl_method="vicreg", stop_gradient=False,
        n_epochs=10, batch_size=4, ema=None, **SSL_OPT,
        backbone_args=dict(spatial_dim=2, in_channels=1, structure=[[8, 8, 5, 2], [16, 16, 3, 2]],
                           maxpool_structure=[[2, 2], [2, 2]], res_type="resnet", adn_fn=adn),
        projection_head_args=dict(in_channels=16, structure=[32, 24], adn_fn=adn1),
        prediction_head_args=dict(in_channels=24, structure=[32, 24], adn_fn=adn1))
    assert [k for k, _ in net.named_parameters()] == list(g["param_keys"])
    assert [",".join(map(str, p.shape)) for _, p in net.named_parameters()] == list(g["param_shapes"])
    net.load_state_dict(fill_state_dict(net.state_dict(), gain=SSL_GAIN))
    net = net.to(cuda).train()
    x1, x2 = torch.from_numpy(g["x1"]).to(cuda), torch.from_numpy(g["x2"]).to(cuda)
    assert rel(net(x1, ret="representation"), g["representation"]) < 1e-4
    # batch norm couples the items of a batch: the two views keep their own passes
    assert net._views_share_a_pass("prediction", "projection") is False
    loss = net.training_step({"a": x1, "b": x2}, 0)
    np.testing.assert_allclose(torch.stack(list(net.last_losses)).detach().cpu().numpy(),
                               g["losses"], rtol=5e-4, atol=1e-6)
    loss.backward()
    for k, p in net.named_parameters():
        assert grad_rel_err(g, k, p.grad.cpu().numpy()) < 5e-3, k
