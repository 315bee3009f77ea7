"""SWIN-UNet (BASELINE config 5 in miniature: convolutional embedding,
8^3 windows of 4^3 patches, shifts [0, 1], anisotropic first stride, conv links) against
fixtures generated from the real reference (oracle/make_golden.py, case swinunet3d_small)."""
import copy
import os

import numpy as np
import pytest
import torch

from adell_mri_amd.modules.activations import activation_factory
from adell_mri_amd.modules.layers.vit import generate_mask
from adell_mri_amd.modules.segmentation.unetr import SWINUNet
from cases import SWIN_CASES
from oracle.torch_ref.unet import compound_loss
from oracle.weights import fill_state_dict

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def build(name="swinunet3d_small", **over):
    kw = copy.deepcopy(SWIN_CASES[name])
    kw.update(over)
    kw["activation_fn"] = activation_factory[kw["activation_fn"]]
    net = SWINUNet(**kw)
    net.load_state_dict(fill_state_dict(net.state_dict()))
    return net


def test_swinunet_state_dict_keys_and_shapes_equal_reference():
    g = np.load(os.path.join(GOLD, "swinunet3d_small.npz"))
    net = build()
    assert [k for k, _ in net.named_parameters()] == [str(k) for k in g["param_keys"]]
    shapes = {str(k): tuple(int(i) for i in str(s).split(",")) for k, s in
              zip(g["param_keys"], g["param_shapes"])}
    for k, p in net.named_parameters():
        assert tuple(p.shape) == shapes[k], k


def test_shift_mask_regions():
    """-100 exactly between tokens whose labels differ; labels follow the reference's strided
    '(w1 h)' reading of each axis (vit.py:95-129, 167-207)."""
    m = generate_mask([8, 8, 4], [2, 2, 2], 1)
    assert m.shape == (4 * 4 * 2, 8, 8)
    assert set(np.unique(m.numpy())) <= {-100.0, 0.0}
    assert torch.equal(m, m.transpose(1, 2)) and float(m.diagonal(dim1=1, dim2=2).abs().max()) == 0
    assert generate_mask([8, 8, 4], [2, 2, 2], 0) is None
    # label image: 27 regions numbered in product order of (0:-w, -w:-s, -s:) per axis
    lab = np.zeros((8, 8, 4))
    sl = [(slice(0, -2), slice(-2, -1), slice(-1, None))] * 3
    c = 0
    for i in range(3):
        for j in range(3):
            for k in range(3):
                lab[sl[0][i], sl[1][j], sl[2][k]] = c
                c += 1
    # window (h, w, d) holds positions (w1 * n_h + h, w2 * n_w + w, w3 * n_d + d)
    win = 1 * 8 + 3 * 2 + 1            # h = 1, w = 3, d = 1 in (4, 4, 2) windows
    toks = [lab[a * 4 + 1, b * 4 + 3, e * 2 + 1] for a in range(2) for b in range(2)
            for e in range(2)]
    want = np.where(np.subtract.outer(toks, toks).T != 0, -100.0, 0.0)
    np.testing.assert_array_equal(m[win].numpy(), want)


def test_swinunet_fails_loudly_without_gpu():
    with pytest.raises(Exception):
        build()(torch.zeros((1, 2, 32, 32, 16)))


@pytest.mark.gpu
def test_swinunet_logits_within_1e4_of_reference(cuda):
    g = np.load(os.path.join(GOLD, "swinunet3d_small.npz"))
    net = build().to(cuda).eval()
    with torch.no_grad():
        logits, _ = net(torch.from_numpy(g["x"]).to(cuda), return_logits=True)
    ref = g["logits"]
    err = np.abs(logits.cpu().numpy() - ref).max() / np.abs(ref).max()
    assert err < 1e-4, err


@pytest.mark.gpu
def test_swinunet_parameter_gradients_match_reference(cuda):
    g = np.load(os.path.join(GOLD, "swinunet3d_small.npz"))
    net = build().to(cuda).eval()
    prob, _ = net(torch.from_numpy(g["x"]).to(cuda))
    loss = compound_loss(prob, torch.from_numpy(g["y"]).to(cuda))
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-4)
    loss.backward()
    for k, p in net.named_parameters():
        if ("grad:" + k) not in g.files:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        # target: the reference's fp64 gradients; the per-voxel 2-channel LayerNorm makes the
        # reference's own fp32 gradients noisy, so the bar is 3e-3 or twice that noise
        ref32, ref64 = g["grad:" + k], g["grad64:" + k]
        scale = np.abs(ref64).max()
        if k.endswith(".bias") and ("grad64:" + k[:-5] + ".weight") in g.files:
            scale = max(scale, 1e-1 * np.abs(g["grad64:" + k[:-5] + ".weight"]).max())
        noise = np.abs(ref32 - ref64).max() / (scale + 1e-12)
        err = np.abs(p.grad.cpu().numpy() - ref64).max() / (scale + 1e-12)
        assert err < max(3e-3, 2 * noise), (k, err, noise)


@pytest.mark.gpu
def test_swinunet_training_mode_with_dropout_runs_and_learns(cuda):
    """dropout_rate > 0 in train(): token dropout, attention dropout and the per-voxel MLP
    dropout all run on the HIP path; a few SGD steps reduce the loss."""
    from adell_mri_amd.optim import FusedSGD

    g = np.load(os.path.join(GOLD, "swinunet3d_small.npz"))
    net = build(dropout_rate=0.1).to(cuda).train()
    opt = FusedSGD(net.parameters(), lr=1e-2, momentum=0.9)
    x, y = torch.from_numpy(g["x"]).to(cuda), torch.from_numpy(g["y"]).to(cuda)
    losses = []
    for _ in range(6):
        opt.zero_grad()
        prob, _ = net(x)
        loss = compound_loss(prob, y)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses


# ---- two dimensions (unetr.py:635-1033 with spatial_dimensions=2; the reference's own tests build it:
# testing/test_swin_unet.py:15, 43-161): depth-1 volumes on the same kernels --------------------------
def test_swinunet2d_state_dict_keys_and_shapes_equal_reference():
    g = np.load(os.path.join(GOLD, "swinunet2d_small.npz"))
    net = build("swinunet2d_small")
    assert [k for k, _ in net.named_parameters()] == [str(k) for k in g["param_keys"]]
    shapes = {str(k): tuple(int(i) for i in str(s).split(",")) for k, s in
              zip(g["param_keys"], g["param_shapes"])}
    for k, p in net.named_parameters():
        assert tuple(p.shape) == shapes[k], k
    m = generate_mask([8, 16], [2, 2], 1)
    assert m.shape == (4 * 8, 4, 4) and set(np.unique(m.numpy())) <= {-100.0, 0.0}


@pytest.mark.gpu
def test_swinunet2d_logits_and_gradients_match_reference(cuda):
    g = np.load(os.path.join(GOLD, "swinunet2d_small.npz"))
    net = build("swinunet2d_small").to(cuda).eval()
    x = torch.from_numpy(g["x"]).to(cuda)
    with torch.no_grad():
        logits, _ = net(x, return_logits=True)
    ref = g["logits"]
    assert logits.shape == ref.shape
    err = np.abs(logits.cpu().numpy() - ref).max() / np.abs(ref).max()
    assert err < 1e-4, err
    prob, _ = net(x)
    loss = compound_loss(prob, torch.from_numpy(g["y"]).to(cuda))
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-4)
    loss.backward()
    for k, p in net.named_parameters():
        if ("grad:" + k) not in g.files:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        ref32, ref64 = g["grad:" + k], g["grad64:" + k]
        scale = np.abs(ref64).max()
        if k.endswith(".bias") and ("grad64:" + k[:-5] + ".weight") in g.files:
            scale = max(scale, 1e-1 * np.abs(g["grad64:" + k[:-5] + ".weight"]).max())
        noise = np.abs(ref32 - ref64).max() / (scale + 1e-12)
        err = np.abs(p.grad.cpu().numpy() - ref64).max() / (scale + 1e-12)
        assert err < max(3e-3, 2 * noise), (k, err, noise)
