"""One training step through each ``*PL`` wrapper as ``get_segmentation_network`` builds it
(-m gpu): UNetPlusPlusPL against the deep-supervision arithmetic of the reference
(pl.py:284-317) evaluated on the reference fixture's own outputs, UNETRPL / SWINUNetPL against
their fixtures' losses."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from adell_mri_amd.modules.config_parsing import parse_config_unet
from adell_mri_amd.trainer import StepRunner
from adell_mri_amd.utils.network_factories import get_segmentation_network
from cases import SWIN_CASES, UNETPP_CASES, UNETR_CASES
from oracle.torch_ref.unet import dice_loss, focal_loss
from oracle.weights import tensor_for

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
LOSS = {"dice": {"eps": 1e-6, "smooth": 1e-5}, "focal": {"gamma": 1.0, "eps": 1e-6}}
TRAIN = dict(learning_rate=5e-4, batch_size=2, weight_decay=5e-3, loss_fn=LOSS)


def _build(net_type, case_kw, n_keys, size=None):
    raw = dict(case_kw)
    for k in ("image_size",):
        raw.pop(k, None)
    raw["in_channels"] = raw["in_channels"] // n_keys
    raw.pop("n_classes")
    cfg, _ = parse_config_unet({**raw, **TRAIN}, n_keys, 2)
    net = get_segmentation_network(
        net_type, cfg, False, [], [], None, None, None, 100, [None], False, None, None, None,
        False, 2, ["image", "image_1"][:n_keys], random_crop_size=size)
    sd = {k: torch.from_numpy(tensor_for(k, v.shape)) if v.numel() > 0 else v
          for k, v in net.state_dict().items()}
    net.load_state_dict(sd)
    return net


def _check_moved(net, before):
    """Every parameter that received a non-zero gradient moved; every parameter without a
    gradient (e.g. the unused gate stacks the factory's feature_conditioning=0 creates) stayed
    exactly where it was (torch.optim semantics)."""
    moved = 0
    for k, p in net.named_parameters():
        same = torch.equal(p.detach().cpu(), before[k])
        if p.grad is None:
            assert same, k
        elif p.numel() and float(p.grad.abs().max()) > 0:
            assert not same, k
            moved += 1
    assert moved > 0.5 * len(before)


def test_unetpp_pl_step_matches_reference_deep_supervision_arithmetic(cuda):
    g = np.load(os.path.join(GOLD, "unetpp3d_small.npz"))
    net = _build("unetpp", UNETPP_CASES["unetpp3d_small"], 2).to(cuda).eval()
    assert type(net).__name__ == "UNetPlusPlusPL" and net.deep_supervision is True
    x, y = torch.from_numpy(g["x"]), torch.from_numpy(g["y"])
    # expected: pl.py:298-316 on the REFERENCE's outputs (prob, aux0, aux1 of the fixture)
    aux = [torch.from_numpy(g[f"aux{i}"]) for i in range(2)]
    t = len(aux)
    def calc(p, tt):   # calculate_loss (pl.py:218-222) for the dice + focal CompoundLoss
        return torch.stack([dice_loss(p, tt).mean(), focal_loss(p, tt).mean()])

    want = calc(torch.from_numpy(g["prob"]), y)
    for i, o in enumerate(aux):
        y_small = (F.interpolate(y, o.shape[-3:], mode="trilinear", align_corners=True) > 0).float()
        want = want + calc(o, y_small).mean() / (2 ** (t - i)) / (t + 1)
    batch = {"image": x.to(cuda), "mask": y.to(cuda)}
    got = net.training_step(batch, 0)
    assert abs(float(got.detach()) - float(want.mean())) < 1e-4 * abs(float(want.mean()))
    # and a full optimiser step through the runner moves every parameter that has a gradient
    before = {k: p.detach().cpu().clone() for k, p in net.named_parameters()}
    loss = StepRunner(net).train_step(batch)
    assert torch.isfinite(loss)
    _check_moved(net, before)


def test_unetr_pl_step(cuda):
    g = np.load(os.path.join(GOLD, "unetr3d_small.npz"))
    kw = dict(UNETR_CASES["unetr3d_small"])
    net = _build("unetr", kw, 1, kw["image_size"]).to(cuda).eval()
    assert type(net).__name__ == "UNETRPL"
    batch = {"image": torch.from_numpy(g["x"]).to(cuda), "mask": torch.from_numpy(g["y"]).to(cuda)}
    got = net.training_step(batch, 0)
    np.testing.assert_allclose(float(got), float(g["loss"]), rtol=1e-4)
    before = {k: p.detach().cpu().clone() for k, p in net.named_parameters()}
    loss = StepRunner(net).train_step(batch)
    assert torch.isfinite(loss)
    _check_moved(net, before)


def test_swin_pl_step(cuda):
    g = np.load(os.path.join(GOLD, "swinunet3d_small.npz"))
    kw = dict(SWIN_CASES["swinunet3d_small"])
    net = _build("swin", kw, 2, kw["image_size"]).to(cuda).eval()
    assert type(net).__name__ == "SWINUNetPL"
    batch = {"image": torch.from_numpy(g["x"]).to(cuda), "mask": torch.from_numpy(g["y"]).to(cuda)}
    got = net.training_step(batch, 0)
    np.testing.assert_allclose(float(got), float(g["loss"]), rtol=1e-4)
    before = {k: p.detach().cpu().clone() for k, p in net.named_parameters()}
    loss = StepRunner(net).train_step(batch)
    assert torch.isfinite(loss)
    _check_moved(net, before)


def test_validation_test_and_predict_steps(cuda):
    """validation_step / test_step = mean of the step loss over micro-batches of the training batch
    size (pl.py:423-524); predict_step takes a batch or one un-batched volume (pl.py:347-373);
    crop_if_necessary centre-crops the target (pl.py:258-282)."""
    from cases import UNET_CASES

    g = np.load(os.path.join(GOLD, "unet3d_cfg2_small.npz"))
    net = _build("unet", UNET_CASES["unet3d_cfg2_small"], 2).to(cuda).eval()
    x, y = torch.from_numpy(g["x"]).to(cuda), torch.from_numpy(g["y"]).to(cuda)
    batch = {"image": x, "mask": y}
    with torch.no_grad():
        net.batch_size, net.train_batch_size = 2, None
        whole = net.validation_step(batch, 0)
        np.testing.assert_allclose(whole.item(), g["loss"], rtol=1e-4)
        net.train_batch_size = 1      # two micro-batches of one item: the mean of their losses
        halves = [net.step(x[i:i + 1], y[i:i + 1], None, None, None)[2].mean() for i in range(2)]
        split = net.test_step(batch, 0)
        np.testing.assert_allclose(split.item(), (halves[0].item() + halves[1].item()) / 2, rtol=1e-5)
        pred = net.predict_step({"image": x}, return_only_segmentation=True)
        np.testing.assert_allclose(pred.cpu().numpy(), g["prob"], rtol=1e-4, atol=1e-6)
        one = net.predict_step({"image": x[1]}, return_only_segmentation=True)
        assert one.shape == pred.shape[1:]
        np.testing.assert_allclose(one.cpu().numpy(), g["prob"][1], rtol=1e-4, atol=1e-6)
    net.make_uniform = True
    yy, pp = net.crop_if_necessary(torch.zeros(1, 1, 10, 11, 12), torch.zeros(1, 1, 8, 8, 12))
    assert tuple(yy.shape) == (1, 1, 8, 8, 12) and pp.shape[2] == 8
