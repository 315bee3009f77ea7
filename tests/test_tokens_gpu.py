"""Token kernels (LayerNorm, Linear-as-conv, attention) and UNETR parity on the GPU."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from adell_mri_amd import functional as HF
from adell_mri_amd.modules.activations import activation_factory
from adell_mri_amd.modules.layers.linear_blocks import MultiHeadSelfAttention
from adell_mri_amd.modules.layers.vit import TransformerBlock
from adell_mri_amd.modules.segmentation.unetr import UNETR
from cases import UNETR_CASES, grad_rel_err
from oracle.torch_ref.unet import compound_loss
from oracle.weights import tensor_for

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


@pytest.mark.parametrize("rows,C", [(37, 64), (216, 512), (5, 4096), (1000, 7), (864, 4096),
                                    (300, 1024)])
def test_layernorm_fwd_bwd_matches_torch_cpu(cuda, rows, C):
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(rows, C, generator=g) * 2 + 1).requires_grad_(True)
    w = torch.randn(C, generator=g).requires_grad_(True)
    b = torch.randn(C, generator=g).requires_grad_(True)
    dy = torch.randn(rows, C, generator=g)
    ref = F.layer_norm(x, (C,), w, b, 1e-5)
    ref.backward(dy)
    xd, wd, bd = (t.detach().to(cuda).requires_grad_(True) for t in (x, w, b))
    out = HF.layer_norm(xd, wd, bd, 1e-5)
    out.backward(dy.to(cuda))
    assert _rel(out.detach().cpu(), ref.detach()) < 1e-5
    assert _rel(xd.grad.cpu(), x.grad) < 5e-5
    assert _rel(wd.grad.cpu(), w.grad) < 5e-5
    assert _rel(bd.grad.cpu(), b.grad) < 5e-5


@pytest.mark.parametrize("lead,bshape", [((2,), (216, 48)), ((1500,), (27, 36)), ((3, 70), (5,)),
                                         ((1,), (64, 8))])
def test_add_bcast_fwd_bwd_matches_torch_cpu(cuda, lead, bshape):
    """Positional-embedding add: few rows (one thread per output) and many rows (chunked
    column sums) take different kernels for the embedding gradient."""
    g = torch.Generator().manual_seed(3)
    a = torch.randn(*lead, *bshape, generator=g).requires_grad_(True)
    b = torch.randn(*bshape, generator=g).requires_grad_(True)
    dy = torch.randn(*lead, *bshape, generator=g)
    (a + b).backward(dy)
    ad, bd = (t.detach().to(cuda).requires_grad_(True) for t in (a, b))
    out = HF.add_bcast(ad, bd)
    out.backward(dy.to(cuda))
    assert torch.equal(out.detach().cpu(), (a + b).detach())
    assert _rel(ad.grad.cpu(), a.grad) == 0.0
    assert _rel(bd.grad.cpu(), b.grad) < 1e-5


@pytest.mark.parametrize("rows,cin,cout", [(216, 512, 1536), (64, 64, 512), (10, 30, 7)])
def test_linear_as_conv_fwd_bwd(cuda, rows, cin, cout):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, rows, cin, generator=g).requires_grad_(True)
    w = (torch.randn(cout, cin, generator=g) / cin ** 0.5).requires_grad_(True)
    b = torch.randn(cout, generator=g).requires_grad_(True)
    r = torch.randn(2, rows, cout, generator=g).requires_grad_(True)
    dy = torch.randn(2, rows, cout, generator=g)
    ref = F.linear(x, w, b) + r
    ref.backward(dy)
    xd, wd, bd, rd = (t.detach().to(cuda).requires_grad_(True) for t in (x, w, b, r))
    out = HF.linear(xd, wd, bd, residual=rd)
    out.backward(dy.to(cuda))
    assert out.shape == ref.shape
    assert _rel(out.detach().cpu(), ref.detach()) < 1e-5
    assert _rel(xd.grad.cpu(), x.grad) < 1e-5
    assert _rel(wd.grad.cpu(), w.grad) < 1e-5
    assert _rel(bd.grad.cpu(), b.grad) < 1e-5
    assert _rel(rd.grad.cpu(), r.grad) < 1e-6


@pytest.mark.parametrize("BH,T,A,Dv,use_bias", [(8, 216, 64, 64, False), (6, 24, 8, 12, True),
                                                 (3, 70, 32, 16, True), (4, 8, 4, 4, False),
                                                 # MFMA path (head dims 32 / 64 / 128), ragged T
                                                 (3, 70, 32, 32, True), (2, 33, 32, 64, False),
                                                 (2, 300, 128, 64, True), (1, 96, 64, 128, False),
                                                 (5, 17, 64, 32, True)])
def test_attention_fwd_bwd_matches_manual_softmax(cuda, BH, T, A, Dv, use_bias):
    """softmax(QK^T/sqrt(d) + bias)V, the identity the reference's own value test checks
    (testing/test_self_attention.py:121-197), here against torch CPU autograd."""
    g = torch.Generator().manual_seed(2)
    q = torch.randn(BH, T, A, generator=g).requires_grad_(True)
    k = torch.randn(BH, T, A, generator=g).requires_grad_(True)
    v = torch.randn(BH, T, Dv, generator=g).requires_grad_(True)
    bias = torch.randn(BH, T, T, generator=g) if use_bias else None
    do = torch.randn(BH, T, Dv, generator=g)
    s = q @ k.transpose(-1, -2) / A ** 0.5
    if bias is not None:
        s = s + bias
    ref = torch.softmax(s, -1) @ v
    ref.backward(do)
    qd, kd, vd = (t.detach().to(cuda).requires_grad_(True) for t in (q, k, v))
    out = HF.attention(qd, kd, vd, None if bias is None else bias.to(cuda))
    out.backward(do.to(cuda))
    assert _rel(out.detach().cpu(), ref.detach()) < 1e-5
    assert _rel(qd.grad.cpu(), q.grad) < 5e-5
    assert _rel(kd.grad.cpu(), k.grad) < 5e-5
    assert _rel(vd.grad.cpu(), v.grad) < 5e-5


@pytest.mark.parametrize("BH,T,A,Dv,use_bias", [(8, 216, 64, 64, False), (3, 70, 16, 24, True),
                                                 (2, 100, 32, 32, True)])
def test_attention_dropout_fwd_bwd_matches_masked_softmax(cuda, BH, T, A, Dv, use_bias):
    """dropout_p of the scaled_dot_product_attention call (linear_blocks.py:407-414): V = identity
    exposes the dropped probabilities, i.e. the mask the kernel drew for (seed, offset); the
    forward and the three gradients with random V then have to match torch CPU autograd on
    (softmax(S) * mask / (1 - p)) V with that same mask."""
    from adell_mri_amd import ops
    g = torch.Generator().manual_seed(3)
    p_drop, seed, offset = 0.1, 1234567891011, 7
    q = torch.randn(BH, T, A, generator=g).requires_grad_(True)
    k = torch.randn(BH, T, A, generator=g).requires_grad_(True)
    v = torch.randn(BH, T, Dv, generator=g).requires_grad_(True)
    bias = torch.randn(BH, T, T, generator=g) if use_bias else None
    do = torch.randn(BH, T, Dv, generator=g)
    scale = 1.0 / A ** 0.5
    bd = None if bias is None else bias.to(cuda)
    qd, kd, vd = (t.detach().to(cuda) for t in (q, k, v))
    eye = torch.eye(T, device=cuda).expand(BH, T, T).contiguous()
    pt, _ = ops.attention_fwd(qd, kd, eye, bd, scale, p_drop, seed, offset)
    p0, _ = ops.attention_fwd(qd, kd, eye, bd, scale)
    kept = pt != 0
    frac = kept.float().mean().item()
    assert abs(frac - (1 - p_drop)) < 0.01, frac
    assert torch.allclose(pt[kept], p0[kept] / (1 - p_drop), rtol=1e-5, atol=1e-8)
    # a different offset draws a different mask
    pt2, _ = ops.attention_fwd(qd, kd, eye, bd, scale, p_drop, seed, offset + 1)
    assert ((pt2 != 0) != kept).float().mean().item() > 0.05
    mask = kept.cpu().float() / (1 - p_drop)
    s = q @ k.transpose(-1, -2) * scale
    if bias is not None:
        s = s + bias
    ref = (torch.softmax(s, -1) * mask) @ v
    ref.backward(do)
    out, lse = ops.attention_fwd(qd, kd, vd, bd, scale, p_drop, seed, offset)
    dq, dk, dv = ops.attention_bwd(qd, kd, vd, bd, out, do.to(cuda), lse, scale, p_drop, seed,
                                   offset)
    assert _rel(out.cpu(), ref.detach()) < 1e-5
    assert _rel(dq.cpu(), q.grad) < 5e-5
    assert _rel(dk.cpu(), k.grad) < 5e-5
    assert _rel(dv.cpu(), v.grad) < 5e-5


def test_mhsa_attention_dropout_trains_beyond_window_size(cuda):
    """unetr.yaml ships dropout_rate 0.1 with 216 tokens: training mode runs, is seeded, and
    eval mode is dropout-free."""
    mha = _load(MultiHeadSelfAttention(32, 32, 48, 32, n_heads=4, dropout_rate=0.25)).to(cuda)
    x = torch.randn(2, 100, 32, generator=torch.Generator().manual_seed(0)).to(cuda)
    mha.train()
    y1 = mha(x)
    y2 = mha(x)
    assert not torch.equal(y1, y2)
    y1.sum().backward()
    assert all(torch.isfinite(p.grad).all() for p in mha.parameters() if p.grad is not None)
    mha.eval()
    with torch.no_grad():
        assert torch.equal(mha(x), mha(x))


def _load(mod):
    mod.load_state_dict({k: torch.from_numpy(tensor_for(k, v.shape))
                         for k, v in mod.state_dict().items()})
    return mod


def test_mhsa_and_transformer_block_match_reference(cuda):
    g = np.load(os.path.join(GOLD, "blocks.npz"))
    x = torch.from_numpy(g["tok_x"]).to(cuda)
    mha = _load(MultiHeadSelfAttention(32, 32, 48, 32, n_heads=4)).to(cuda).eval()
    with torch.no_grad():
        y = mha(x)
    np.testing.assert_allclose(y.cpu().numpy(), g["mha_y"], rtol=1e-4, atol=1e-5)
    tb = _load(TransformerBlock(32, 32, 32, n_heads=4, mlp_structure=[64])).to(cuda).eval()
    with torch.no_grad():
        y = tb(x)
    np.testing.assert_allclose(y.cpu().numpy(), g["tb_y"], rtol=1e-4, atol=1e-5)


UNETR_NAMES = ["unetr3d_small", "unetr2d_small"]


def _build_unetr(device, name="unetr3d_small"):
    kw = dict(UNETR_CASES[name])
    kw["activation_fn"] = activation_factory[kw["activation_fn"]]
    return _load(UNETR(**kw)).to(device)


@pytest.mark.parametrize("name", UNETR_NAMES)
def test_unetr_logits_within_1e4_of_reference(cuda, name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    net = _build_unetr(cuda, name).eval()
    with torch.no_grad():
        logits, _ = net(torch.from_numpy(g["x"]).to(cuda), return_logits=True)
    ref = g["logits"]
    rel = np.abs(logits.cpu().numpy() - ref).max() / np.abs(ref).max()
    assert rel < 1e-4, rel


@pytest.mark.parametrize("name", UNETR_NAMES)
def test_unetr_parameter_gradients_match_reference(cuda, name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    net = _build_unetr(cuda, name).eval()
    prob, _ = net(torch.from_numpy(g["x"]).to(cuda))
    loss = compound_loss(prob, torch.from_numpy(g["y"]).to(cuda))
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-4)
    loss.backward()
    for k, p in net.named_parameters():
        if ("grad:" + k) not in g.files:  # untouched by the forward in the reference too
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        err = grad_rel_err(g, k, p.grad.cpu().numpy())
        assert err < 2e-3, (k, err)


@pytest.mark.parametrize("BH,T,A,Dv", [(4, 216, 64, 64), (2, 77, 32, 128)])
def test_mfma_attention_equals_vector_alu_attention(cuda, BH, T, A, Dv):
    """The fp32-MFMA kernels (csrc/tokens.hip, adell_attn_mfma_*) against the vector-ALU kernels
    they replace for head dims 32 / 64 / 128: same outputs, log-sum-exp and gradients, with
    attention dropout drawing the same mask."""
    from adell_mri_amd import _lib, ops
    g = torch.Generator().manual_seed(BH + T)
    q, k = (torch.randn(BH, T, A, generator=g).to(cuda) for _ in range(2))
    v, do = (torch.randn(BH, T, Dv, generator=g).to(cuda) for _ in range(2))
    scale = 1.0 / A ** 0.5
    res = {}
    for name, sw in (("mfma", 0), ("valu", 1)):
        with _lib.tuning(attn_nomfma=sw):
            out, lse = ops.attention_fwd(q, k, v, None, scale, 0.1, 99, 3)
            res[name] = (out, lse, *ops.attention_bwd(q, k, v, None, out, do, lse, scale, 0.1, 99, 3))
    for a, b in zip(res["mfma"], res["valu"]):
        assert _rel(a, b) < 2e-5
