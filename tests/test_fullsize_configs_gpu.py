"""BASELINE configs 3, 4 and 5 at their FULL sizes (-m gpu): built from the YAMLs in configs/
through the same factories the entrypoints use, one training step each, with the
size-independent properties the domain offers -- probabilities in (0, 1), finite loss and
gradients, every gradient-carrying parameter moved by the optimiser step, eval-mode determinism,
batch items independent of one another (instance / layer norms only), the parameter counts the
reference reports for these configurations."""
import os

import pytest
import torch

from adell_mri_amd.modules.config_parsing import parse_config_ssl, parse_config_unet
from adell_mri_amd.trainer import StepRunner
from adell_mri_amd.utils.network_factories import get_segmentation_network, get_ssl_network

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONFIGS = os.path.join(ROOT, "configs")


def _seg(net_type, yaml_name, keys, size, patch=None):
    cfg, _ = parse_config_unet(os.path.join(CONFIGS, yaml_name), len(keys), 2)
    if patch is not None:
        cfg["patch_size"] = patch
    torch.manual_seed(0)
    return get_segmentation_network(
        net_type, cfg, False, [], [], None, None, None, 100, [None], False, None, None, None,
        False, 2, keys, random_crop_size=size)


def _batch(n, c, size, cuda, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand((n, c, *size), generator=g).to(cuda)
    y = (torch.rand((n, 1, *size), generator=g) > 0.9).float().to(cuda)
    return {"image": x, "mask": y}


def _step_properties(net, batch, expect_params):
    n_params = sum(p.numel() for p in net.parameters())
    assert n_params == expect_params, n_params
    net.eval()
    with torch.no_grad():
        p1 = net(batch["image"])[0]
        p2 = net(batch["image"])[0]
        one = net(batch["image"][:1])[0]
    assert torch.equal(p1, p2)                                   # fixed-order reductions
    assert float(p1.min()) > 0.0 and float(p1.max()) < 1.0       # sigmoid head
    scale = float(p1.max() - p1.min())
    assert float((one - p1[:1]).abs().max()) < 1e-5 * max(scale, 1e-3)   # items independent
    net.train()
    before = {k: p.detach().clone() for k, p in net.named_parameters()}
    loss = StepRunner(net).train_step(batch)
    assert torch.isfinite(loss)
    for k, p in net.named_parameters():
        if p.grad is None:
            assert torch.equal(p.detach(), before[k]), k
            continue
        assert torch.isfinite(p.grad).all(), k
        if float(p.grad.abs().max()) > 0:
            assert not torch.equal(p.detach(), before[k]), k
    return float(loss.detach())


def test_config3_unetr_full_size(cuda):
    """unetr.yaml at image 96^3 / patch 16^3 (SURVEY.md 8(d) cfg 3), batch 2. Through the factory
    the model also carries the never-used feature_conditioning = 0 gate stacks."""
    net = _seg("unetr", "unetr.yaml", ["image"], [96, 96, 96], patch=[16, 16, 16]).to(cuda)
    gates = sum(p.numel() for k, p in net.named_parameters() if k.startswith("feature_conditioning_ops"))
    assert sum(p.numel() for p in net.parameters()) - gates == 36068253
    loss = _step_properties(net, _batch(2, 1, (96, 96, 96), cuda), 36068253 + gates)
    assert 0.0 < loss < 5.0


def test_config5_swin_unet_full_size(cuda):
    """unet-swin.yaml at 256 x 256 x 128, 2 channels, batch 1 (cfg 5)."""
    net = _seg("swin", "unet-swin.yaml", ["image", "image_1"], [256, 256, 128]).to(cuda)
    gates = sum(p.numel() for k, p in net.named_parameters() if k.startswith("feature_conditioning_ops"))
    assert sum(p.numel() for p in net.parameters()) - gates == 12626945
    net.train()
    batch = _batch(1, 2, (256, 256, 128), cuda)
    before = {k: p.detach().clone() for k, p in net.named_parameters()}
    loss = StepRunner(net).train_step(batch)
    assert torch.isfinite(loss) and 0.0 < float(loss.detach()) < 5.0
    moved = sum(not torch.equal(before[k], p.detach()) for k, p in net.named_parameters()
                if p.grad is not None)
    assert moved > 200
    net.eval()
    with torch.no_grad():
        p1 = net(batch["image"])[0]
    assert p1.shape == (1, 1, 256, 256, 128) and float(p1.min()) > 0.0 and float(p1.max()) < 1.0


def test_config4_vicreg_convnext_full_size(cuda):
    """ssl-3d-convnext.yaml (cfg 4): two views of 32 crops of 64^3 (B = 32 per GPU),
    VICReg, AdamW."""
    _, cfg = parse_config_ssl(os.path.join(CONFIGS, "ssl-3d-convnext.yaml"), 0.0, 1)
    cfg.pop("batch_size", None)
    cfg["vic_reg_loss_params"] = {}
    torch.manual_seed(0)
    from adell_mri_amd.modules.self_supervised.pl import SelfSLConvNeXtPL
    cfg["backbone_args"] = {k: v for k, v in cfg["backbone_args"].items() if k != "res_type"}
    net = SelfSLConvNeXtPL(aug_image_key_1="augmented_image_1", aug_image_key_2="augmented_image_2",
                           ssl_method="vicreg", stop_gradient=False, n_epochs=100, **cfg).to(cuda)
    assert sum(p.numel() for p in net.parameters()) == 33862368
    assert (net.learning_rate, net.weight_decay) == (0.005, 0.001)
    g = torch.Generator().manual_seed(1)
    x1 = torch.randn((32, 1, 64, 64, 64), generator=g).to(cuda)
    x2 = (x1 + 0.3 * torch.randn(x1.shape, generator=g).to(cuda)).flip(2)
    batch = {"augmented_image_1": x1, "augmented_image_2": x2}
    net.train()
    before = {k: p.detach().clone() for k, p in net.named_parameters()}
    loss = StepRunner(net).train_step(batch)
    assert torch.isfinite(loss)
    assert len(net.last_losses) == 3 and all(torch.isfinite(t) for t in net.last_losses)
    moved = sum(not torch.equal(before[k], p.detach()) for k, p in net.named_parameters()
                if p.grad is not None)
    assert moved > 0.9 * len(before)
    # get_ssl_network routes "vicreg" to the ResNet wrapper (network_factories.py:749-796)
    small = {"backbone_args": dict(spatial_dim=3, in_channels=1, structure=[[8, 8, 3, 1]],
                                   maxpool_structure=[[2, 2, 2]], res_type="resnet",
                                   adn_fn=torch.nn.Identity),
             "projection_head_args": dict(in_channels=8, structure=[16, 8], adn_fn=torch.nn.Identity),
             "prediction_head_args": dict(in_channels=8, structure=[16, 8], adn_fn=torch.nn.Identity)}
    r = get_ssl_network(None, 10, 100, 0, "vicreg", None, "resnet", small, False).to(cuda).train()
    xb = {"aug_image_1": torch.randn(6, 1, 16, 16, 16, device=cuda),
          "aug_image_2": torch.randn(6, 1, 16, 16, 16, device=cuda)}
    assert torch.isfinite(StepRunner(r).train_step(xb))


# ---- round 4: VALUES at full size against the REAL reference ------------------------------------------
# tests/golden/{unetr_cfg3_full, convnext_cfg4_full, swinunet_cfg5_full}.npz were written by
# `python oracle/make_golden.py fullseg` / `fullssl` from the imported reference classes (name-keyed
# weights, seeded inputs regenerated here): logit statistics, 64 sampled voxels, a line and the eight
# corners, the loss terms, and for every parameter gradient its L2 norm, maximum and 16 sampled entries.
import numpy as np  # noqa: E402

from oracle.fullsize import full_inputs  # noqa: E402
from oracle.weights import fill_state_dict  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def _check_logits(g, logits, tag, bar=1e-4):
    lg = logits.detach().float().cpu()
    stats = g["logit_stats"]
    rng = float(stats[3] - stats[2])
    flat = lg.reshape(-1)
    got = [float(lg.double().mean()), float(lg.double().std()), float(lg.min()), float(lg.max()),
           float(lg.double().abs().mean())]
    err = {"sampled": float(np.abs(flat[g["logit_pos"]].numpy() - g["logit_val"]).max()) / rng,
           "line": float(np.abs(lg[0, 0, lg.shape[2] // 2, lg.shape[3] // 2, :].numpy()
                                - g["logit_line"]).max()) / rng,
           "corners": float(np.abs(lg[0, 0][::lg.shape[2] - 1, ::lg.shape[3] - 1,
                                            ::lg.shape[4] - 1].numpy() - g["logit_corners"]).max()) / rng,
           "stats": float(np.abs(np.array(got) - stats).max()) / rng}
    print(f"{tag}: full-size logits vs reference, error / logit range: {err}")
    assert max(err.values()) < bar, err


def _check_grads(g, net, tag, norm_bar=2e-3, entry_bar=3e-3):
    top = max(float(g["gnorm:" + str(k)][1]) for k in g["grad_keys"])
    worst = (0.0, "")
    for k, p in net.named_parameters():
        if ("gnorm:" + k) not in g.files:
            continue
        assert p.grad is not None, k
        ref_norm, ref_max = (float(v) for v in g["gnorm:" + k])
        scale = max(ref_max, 5e-4 * top)
        if k.endswith(".bias") and ("gnorm:" + k[:-5] + ".weight") in g.files:
            # (a bias in front of a norm has a mathematically zero gradient: both sides hold the
            # rounding noise of a sum over ~10^6 voxels, measured against the sibling weight's scale)
            scale = max(scale, float(g["gnorm:" + k[:-5] + ".weight"][1]))
        got = p.grad.detach().reshape(-1)
        e = float(np.abs(got[g["gpos:" + k]].cpu().numpy() - g["gval:" + k]).max()) / scale
        worst = max(worst, (e, k))
        assert e < entry_bar, (k, e)
        n = float(got.double().norm())
        # (norm of a gradient that is mathematically zero -- a bias in front of a norm -- is the norm
        # of rounding noise: the bar is relative plus the entry-level floor over the tensor)
        assert abs(n - ref_norm) <= norm_bar * ref_norm + entry_bar * scale * np.sqrt(got.numel()), \
            (k, n, ref_norm)
    print(f"{tag}: worst sampled gradient entry error / scale {worst}")


def test_config3_unetr_values_match_reference(cuda):
    g = np.load(os.path.join(GOLD, "unetr_cfg3_full.npz"))
    shape = tuple(int(v) for v in g["shape"])
    x, y = full_inputs(shape, int(g["seed"]))
    assert abs(float(x.double().sum()) - g["x_checksum"][0]) < 1e-6 * g["x_checksum"][0]
    net = _seg("unetr", "unetr.yaml", ["image"], [96, 96, 96], patch=[16, 16, 16])
    net.load_state_dict(fill_state_dict(net.state_dict()))
    assert [k for k, _ in net.named_parameters()] == [str(k) for k in g["param_keys"]]
    net = net.to(cuda).eval()
    logits = net(x.to(cuda), return_logits=True)[0]
    _check_logits(g, logits, "cfg3 UNETR 96^3")
    from oracle.torch_ref.unet import compound_loss
    prob = net(x.to(cuda))[0]
    loss = compound_loss(prob, y.to(cuda))
    np.testing.assert_allclose(float(loss.detach()), float(g["loss"]), rtol=1e-4)
    loss.backward()
    _check_grads(g, net, "cfg3 UNETR 96^3")


def test_config5_swinunet_forward_values_match_reference(cuda):
    g = np.load(os.path.join(GOLD, "swinunet_cfg5_full.npz"))
    shape = tuple(int(v) for v in g["shape"])
    x, _ = full_inputs(shape, int(g["seed"]))
    assert abs(float(x.double().sum()) - g["x_checksum"][0]) < 1e-6 * g["x_checksum"][0]
    net = _seg("swin", "unet-swin.yaml", ["image", "image_1"], [256, 256, 128])
    net.load_state_dict(fill_state_dict(net.state_dict()))
    assert [k for k, _ in net.named_parameters()] == [str(k) for k in g["param_keys"]]
    net = net.to(cuda).eval()
    with torch.no_grad():
        logits = net(x.to(cuda), return_logits=True)[0]
    _check_logits(g, logits, "cfg5 SWIN-UNet 256x256x128")


def test_config4_convnext_values_match_reference(cuda):
    from adell_mri_amd.modules.layers.adn_fn import get_adn_fn
    from adell_mri_amd.modules.layers.conv_next import ConvNeXt
    from adell_mri_amd.modules.self_supervised.losses.vicreg import VICRegLoss
    import yaml

    g = np.load(os.path.join(GOLD, "convnext_cfg4_full.npz"))
    with open(os.path.join(CONFIGS, "ssl-3d-convnext.yaml")) as fh:
        cfg = yaml.safe_load(fh)
    adn3 = get_adn_fn(3, cfg["norm_fn"], cfg["act_fn"], 0.0)
    adn1 = get_adn_fn(1, cfg["norm_fn"], cfg["act_fn"], 0.0)
    bb = {k: v for k, v in cfg["backbone_args"].items() if k != "res_type"}
    bb["adn_fn"] = adn3
    net = ConvNeXt(backbone_args=bb,
                   projection_head_args=dict(cfg["projection_head_args"], adn_fn=adn1),
                   prediction_head_args=dict(cfg["prediction_head_args"], adn_fn=adn1))
    net.load_state_dict(fill_state_dict(net.state_dict(), gain=3.0, norm_weight_offset=1.0))
    assert [k for k, _ in net.named_parameters()] == [str(k) for k in g["param_keys"]]
    # a non-collapsed state: all three VICReg terms carry weight (the covariance term was 5e-8)
    assert g["losses"][2] > 1e-2 and float(np.std(g["y1"], axis=0).mean()) > 0.05
    net = net.to(cuda).train()
    zz, yy, xx = torch.meshgrid(*[torch.arange(64.0)] * 3, indexing="ij")
    gen = torch.Generator().manual_seed(20260128)
    x1 = torch.stack([torch.sin((b + 1) * 0.21 * zz) * torch.cos((b + 2) * 0.13 * yy)
                      + 0.02 * (b - 1.5) * xx for b in range(4)])[:, None]
    x1 = x1 + 0.2 * torch.rand(x1.shape, generator=gen)
    x2 = (x1 + 0.3 * torch.randn(x1.shape, generator=gen)).flip(2)
    assert abs(float(x1.double().sum()) - g["x_checksum"][0]) < 1e-6 * abs(g["x_checksum"][0])
    x1, x2 = x1.to(cuda), x2.to(cuda)
    with torch.no_grad():
        rep = net(x1, ret="representation")
    scale = float(np.abs(g["representation"]).max())
    assert float(np.abs(rep.cpu().numpy() - g["representation"]).max()) < 1e-4 * scale
    y1, y2 = net(x1, ret="prediction"), net(x2, ret="projection")
    for got, key in ((y1, "y1"), (y2, "y2")):
        s = float(np.abs(g[key]).max())
        assert float(np.abs(got.detach().cpu().numpy() - g[key]).max()) < 2e-4 * s, key
    losses = VICRegLoss()(y1, y2)
    np.testing.assert_allclose(torch.stack(list(losses)).detach().cpu().numpy(), g["losses"],
                               rtol=2e-3, atol=1e-6)
    sum(losses).backward()
    _check_grads(g, net, "cfg4 ConvNeXt 64^3", norm_bar=5e-3, entry_bar=5e-3)


def test_config2b_resnet_backbone_values_match_reference(cuda):
    """BASELINE config 2b at 1 x 2 x 128^3 ("measure it second"): the U-Net with the
    ResNet-backbone encoder, built through the factory path the entrypoint takes (bench.build_cfg2b:
    parse_config_ssl -> handoff -> get_segmentation_network), against tests/golden/unet3d_cfg2b_full.npz
    (`python oracle/make_golden.py full unet3d_cfg2b_full`: the REAL reference classes assembled as
    train.py:672-734 assembles them): logits within 1e-4 of the logit range on the odd-sized maps whose
    crop_to_size path is live (65^3 / 33x33x65 / 17x17x65 / 9x9x33), loss, every parameter gradient."""
    import bench

    g = np.load(os.path.join(GOLD, "unet3d_cfg2b_full.npz"))
    shape = tuple(int(v) for v in g["shape"])
    x, y = full_inputs(shape, int(g["seed"]))
    assert abs(float(x.double().sum()) - g["x_checksum"][0]) < 1e-6 * g["x_checksum"][0]
    net, _ = bench.build_cfg2b()
    assert sum(p.numel() for p in net.parameters()) == 41819841
    net.load_state_dict(fill_state_dict(net.state_dict()))
    assert [k for k, _ in net.named_parameters()] == [str(k) for k in g["param_keys"]]
    net = net.to(cuda).eval()
    logits = net(x.to(cuda), return_logits=True)[0]
    _check_logits(g, logits, "cfg2b ResNet-backbone U-Net 128^3")
    from oracle.torch_ref.unet import compound_loss
    prob = net(x.to(cuda))[0]
    loss = compound_loss(prob, y.to(cuda))
    np.testing.assert_allclose(float(loss.detach()), float(g["loss"]), rtol=1e-4)
    loss.backward()
    _check_grads(g, net, "cfg2b ResNet-backbone U-Net 128^3")
