"""``link_type="attention"`` (unet.py:473-481: SelfAttentionBlock over [16, 16, 1] patches of every
skip tensor, self_attention.py:152-239; the reference tests it for shapes only,
testing/test_unet.py:204-235) against fixtures generated from the real reference
(`python oracle/make_golden.py attention`): logits 1e-4, loss, every parameter gradient.

(The 2-D case uses a smooth activation on purpose: with ReLU its first draft had ONE bottleneck
pre-activation within 2e-7 of zero, whose mask flipped between the f16x3 and the fp32 forward --
at 2 x 16 x 24 bottleneck voxels a single flipped element moves the encoder's weight gradients by
~1 %, in either direction, reference included; traced with tools/dbg_att.py, round 4.)"""
import os

import numpy as np
import pytest
import torch

from adell_mri_amd.modules.activations import activation_factory
from adell_mri_amd.modules.segmentation.unet import UNet
from cases import ATTENTION_LINK_CASES
from oracle.torch_ref.unet import compound_loss
from oracle.weights import tensor_for

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def build(name):
    kw = dict(ATTENTION_LINK_CASES[name])
    kw["activation_fn"] = activation_factory[kw["activation_fn"]]
    net = UNet(**kw)
    net.load_state_dict({k: torch.from_numpy(tensor_for(k, v.shape))
                         for k, v in net.state_dict().items()})
    return net


@pytest.mark.parametrize("name", list(ATTENTION_LINK_CASES))
def test_parameter_tree_equals_reference(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    net = build(name)
    assert [k for k, _ in net.named_parameters()] == [str(k) for k in g["param_keys"]]
    shapes = {str(k): tuple(int(i) for i in str(s).split(",")) for k, s in
              zip(g["param_keys"], g["param_shapes"])}
    for k, p in net.named_parameters():
        assert tuple(p.shape) == shapes[k], k


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(ATTENTION_LINK_CASES))
def test_logits_and_gradients_match_reference(cuda, name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    net = build(name).to(cuda).eval()
    x = torch.from_numpy(g["x"]).to(cuda)
    with torch.no_grad():
        logits, _ = net(x, return_logits=True)
    ref = g["logits"]
    assert np.abs(logits.cpu().numpy() - ref).max() / np.abs(ref).max() < 1e-4
    prob, _ = net(x)
    loss = compound_loss(prob, torch.from_numpy(g["y"]).to(cuda))
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-4)
    loss.backward()
    for k, p in net.named_parameters():
        if ("grad:" + k) not in g.files:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        # target: the reference's fp64 gradients. QK-LayerNorm over head dimensions of 2-8 makes the
        # reference's own fp32 gradients noisy, so the bar is 3e-3 or twice that noise (as for the
        # SWIN blocks, tests/test_swin.py)
        ref32, ref64 = g["grad:" + k], g["grad64:" + k]
        scale = np.abs(ref64).max()
        if k.endswith(".bias") and ("grad64:" + k[:-5] + ".weight") in g.files:
            scale = max(scale, 1e-1 * np.abs(g["grad64:" + k[:-5] + ".weight"]).max())
        noise = np.abs(ref32 - ref64).max() / (scale + 1e-12)
        err = np.abs(p.grad.cpu().numpy() - ref64).max() / (scale + 1e-12)
        assert err < max(3e-3, 2 * noise), (k, err, noise)
