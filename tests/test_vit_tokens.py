"""LinearEmbedding / ViT options of adell_mri/modules/layers/vit.py:389-881, 1731-1794 that UNETR does
not use -- class token, registers, the fixed sinusoidal table, patch erasing -- against outputs and
gradients of the REAL reference (tests/golden/vit_tokens.npz, oracle/make_golden.py vit)."""
import os

import numpy as np
import pytest
import torch

from oracle.weights import tensor_for

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vit_tokens.npz")
KW = dict(image_size=[16, 16, 16], patch_size=[8, 8, 8], in_channels=2, number_of_blocks=2,
          attention_dim=32, hidden_dim=32, embedding_size=32, n_heads=4, dropout_rate=0.0,
          mlp_structure=[64])


def _vit(**extra):
    from adell_mri_amd.modules.layers.adn_fn import get_adn_fn
    from adell_mri_amd.modules.layers.vit import ViT

    return ViT(**KW, adn_fn=get_adn_fn(1, "identity", "gelu", 0.0), **extra)


def _load(net):
    sd = {k: torch.from_numpy(tensor_for(k, v.shape))
          for k, v in net.state_dict().items() if v.is_floating_point() and v.numel() > 0}
    net.load_state_dict(sd, strict=False)
    return net


def test_parameter_tree_and_sinusoidal_table_match_the_reference():
    g = np.load(GOLD)
    net = _vit(use_class_token=True, n_registers=2, learnable_embedding=False)
    assert list(net.state_dict().keys()) == list(g["state_keys"])
    table = net.embedding.positional_embedding
    assert not table.requires_grad
    np.testing.assert_allclose(table.detach().numpy(), g["pos_init"], rtol=0, atol=1e-7)
    assert net.embedding.class_token.shape == (1, 1, 32) and net.embedding.registers.shape == (1, 2, 32)


def _close(a, b, tol, what):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else a
    err = np.abs(a - b).max()
    assert err <= tol * max(np.abs(b).max(), 1e-6), (what, err, np.abs(b).max())


@pytest.mark.gpu
def test_class_token_registers_sinusoidal_forward_and_gradients(cuda):
    g = np.load(GOLD)
    net = _load(_vit(use_class_token=True, n_registers=2, learnable_embedding=False)).to(cuda).eval()
    x = torch.from_numpy(g["x"]).to(cuda)
    y, hidden = net(x, return_at=[0])
    assert tuple(y.shape) == (2, 11, 32)            # 2 registers + class token + 8 patches
    _close(y, g["y"], 1e-4, "y")
    _close(hidden[0], g["hidden0"], 1e-4, "hidden0")
    (y * torch.from_numpy(g["wgt"]).to(cuda)).sum().backward()
    grads = dict(net.named_parameters())
    keys = list(g["grad_keys"])
    assert "embedding.class_token" in keys and "embedding.registers" in keys
    assert "embedding.positional_embedding" not in keys
    scale = max(np.abs(g["grad:" + k]).max() for k in keys)
    for k in keys:
        ref = g["grad:" + k]
        got = grads[k].grad
        assert got is not None, k
        err = np.abs(got.cpu().numpy() - ref).max()
        assert err <= 2e-4 * max(np.abs(ref).max(), 1e-3 * scale), (k, err)
    assert grads["embedding.positional_embedding"].grad is None


@pytest.mark.gpu
def test_patch_erasing_draws_the_reference_mask(cuda):
    g = np.load(GOLD)
    net = _load(_vit(use_class_token=True, patch_erasing=0.4)).to(cuda).train()
    x = torch.from_numpy(g["x"]).to(cuda)
    torch.manual_seed(5)
    y, _ = net(x)
    _close(y, g["y_erased"], 1e-4, "erased")
    # and the op by itself: whole tokens zeroed, nothing rescaled
    from adell_mri_amd.modules.layers.regularization import ChannelDropout

    op = ChannelDropout(0.4).train()
    t = torch.randn(2, 9, 32, device=cuda) + 3.0
    torch.manual_seed(5)
    z = op(t)
    keep = torch.from_numpy(g["erase_mask"]).to(cuda)
    assert torch.equal(z != 0, keep[:, :, None].expand_as(z))
    assert torch.equal(z[keep], t[keep])
    assert op.eval()(t) is t


def test_channel_tokens_parameter_tree_matches_the_reference():
    g = np.load(GOLD)
    net = _vit(channel_to_token=True)
    assert list(net.state_dict().keys()) == list(g["c2t_state_keys"])
    # 8 patches x 2 channels = 16 tokens of 8^3 features
    assert net.embedding.n_patches == 16 and net.embedding.n_features == 512


@pytest.mark.gpu
def test_channel_tokens_forward_and_gradients(cuda):
    """LinearEmbedding(channel_to_token=True) (vit.py:424-467, 566-571, 622-645): every channel of
    a patch is a token, ordered (h w c d) as the reference's rearrangement orders them."""
    g = np.load(GOLD)
    net = _load(_vit(channel_to_token=True)).to(cuda).eval()
    x = torch.from_numpy(g["x"]).to(cuda)
    y, _ = net(x)
    assert tuple(y.shape) == (2, 16, 32)
    _close(y, g["c2t_y"], 1e-4, "y")
    (y * torch.from_numpy(g["c2t_w"]).to(cuda)).sum().backward()
    grads = dict(net.named_parameters())
    keys = [str(k) for k in g["c2t_grad_keys"]]
    scale = max(np.abs(g["c2t_grad:" + k]).max() for k in keys)
    for k in keys:       # (a k-norm bias has a mathematically zero gradient: floor by the largest)
        ref = g["c2t_grad:" + k]
        err = np.abs(grads[k].grad.cpu().numpy() - ref).max()
        assert err <= 2e-4 * max(np.abs(ref).max(), 1e-3 * scale), (k, err)
