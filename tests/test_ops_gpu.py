"""Op-level parity of the HIP kernels against the plain-C oracle (oracle/c),
called through the C ABI. Tolerances are written per test; fp32 everywhere."""
import numpy as np
import pytest
import torch

from adell_mri_amd import _lib, ops
from oracle import cops

pytestmark = pytest.mark.gpu


def _dev(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def _cl(a, device):
    """numpy NCDHW -> cuda tensor, logical NCDHW, NDHWC memory."""
    return ops.ndhwc(_dev(a, device))


def _np(t):
    return t.detach().contiguous().cpu().numpy()


def _relerr(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


CONV_CASES = [
    # N, Cin, size, Cout, k, s, p, cfg
    (1, 32, (16, 16, 16), 32, 3, 1, 1, -1),
    (2, 2, (12, 10, 9), 32, 3, 1, 1, -1),
    (1, 2, (9, 9, 9), 2, 3, 1, 1, -1),
    (1, 64, (8, 8, 8), 64, 3, 1, 1, -1),
    (1, 32, (16, 16, 16), 32, 3, 2, 1, -1),
    (1, 16, (17, 15, 13), 48, 3, 2, 1, -1),
    (1, 32, (8, 8, 8), 1, 1, 1, 0, -1),
    (1, 40, (8, 8, 8), 72, 3, 1, 1, -1),
    (1, 32, (16, 16, 16), 64, 3, 1, 1, 0),
    (1, 32, (16, 16, 16), 32, 3, 1, 1, 1),
    (1, 32, (16, 16, 16), 64, 3, 1, 1, 2),
    (1, 32, (16, 16, 16), 32, 3, 1, 1, 3),
    (1, 8, (6, 6, 6), 8, (3, 3, 1), (2, 2, 1), (1, 1, 0), -1),
    (1, 8, (10, 10, 10), 8, 3, 1, 0, -1),
]


@pytest.mark.parametrize("N,Cin,size,Cout,k,s,p,cfg", CONV_CASES)
def test_conv3d_fwd_matches_oracle(cuda, N, Cin, size, Cout, k, s, p, cfg):
    rng = np.random.default_rng(1234)
    k3 = ops._triple(k)
    x = rng.standard_normal((N, Cin, *size)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, *k3)) / np.sqrt(Cin * np.prod(k3))).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    ref = cops.conv3d(x, w, b, s, p)
    _lib.lib().adell_debug_force_conv_cfg(cfg)
    try:
        wp = ops.pack_weight(_dev(w, cuda), 0)
        y, part = ops.conv3d_fwd(_cl(x, cuda), wp, _dev(b, cuda), Cout, k, s, p, want_stats=True)
        torch.cuda.synchronize()
    finally:
        _lib.lib().adell_debug_force_conv_cfg(-1)
    got = _np(y)
    assert got.shape == ref.shape
    # fp32 MFMA k-ordered fma chain vs fp64-accumulated oracle
    assert _relerr(got, ref) < 2e-6 * np.sqrt(Cin * np.prod(k3)) + 1e-6
    # fused statistics epilogue
    V = np.prod(ref.shape[2:])
    mean, rstd = ops.stats_finalize(part, V, 1e-5)
    m_ref = ref.reshape(N, Cout, -1).mean(-1)
    r_ref = 1.0 / np.sqrt(ref.reshape(N, Cout, -1).var(-1) + 1e-5)
    np.testing.assert_allclose(_np(mean), m_ref, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(_np(rstd), r_ref, rtol=1e-4)


def test_conv3d_virtual_concat_and_residual(cuda):
    rng = np.random.default_rng(7)
    xa = rng.standard_normal((1, 32, 8, 8, 8)).astype(np.float32)
    xb = rng.standard_normal((1, 32, 8, 8, 8)).astype(np.float32)
    w = (rng.standard_normal((64, 64, 3, 3, 3)) * 0.03).astype(np.float32)
    b = rng.standard_normal(64).astype(np.float32)
    res = rng.standard_normal((1, 64, 8, 8, 8)).astype(np.float32)
    ref = cops.conv3d(np.concatenate([xa, xb], 1), w, b, 1, 1) + res
    wp = ops.pack_weight(_dev(w, cuda), 0)
    y, _ = ops.conv3d_fwd(_cl(xa, cuda), wp, _dev(b, cuda), 64, 3, 1, 1, x1=_cl(xb, cuda),
                          residual=_cl(res, cuda))
    assert _relerr(_np(y), ref) < 1e-4


BWD_CASES = [
    (1, 32, (8, 8, 8), 32, 3, 1, 1),
    (1, 16, (16, 16, 16), 24, 3, 2, 1),
    (1, 8, (9, 9, 9), 8, 3, 2, 1),
    (2, 2, (8, 8, 8), 32, 3, 1, 1),
    (1, 32, (8, 8, 8), 1, 1, 1, 0),
]


@pytest.mark.parametrize("N,Cin,size,Cout,k,s,p", BWD_CASES)
def test_conv3d_bwd_data_matches_oracle(cuda, N, Cin, size, Cout, k, s, p):
    rng = np.random.default_rng(99)
    x = rng.standard_normal((N, Cin, *size)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, k, k, k)) / np.sqrt(Cout * k ** 3)).astype(np.float32)
    osz = ops.conv_out_size(size, (k,) * 3, (s,) * 3, (p,) * 3)
    dy = rng.standard_normal((N, Cout, *osz)).astype(np.float32)
    dx_ref, _, _ = cops.conv3d_bwd(x, w, dy, s, p)
    wpb = ops.pack_weight(_dev(w, cuda), 1)
    dx0, dx1 = ops.conv3d_bwd_data(_cl(dy, cuda), wpb, size, Cin, 0, k, s, p)
    assert dx1 is None
    assert _relerr(_np(dx0), dx_ref) < 1e-4


def test_conv3d_bwd_data_splits_concat_sources(cuda):
    rng = np.random.default_rng(5)
    x = rng.standard_normal((1, 48, 8, 8, 8)).astype(np.float32)
    w = (rng.standard_normal((32, 48, 3, 3, 3)) * 0.03).astype(np.float32)
    dy = rng.standard_normal((1, 32, 8, 8, 8)).astype(np.float32)
    dx_ref, _, _ = cops.conv3d_bwd(x, w, dy, 1, 1)
    wpb = ops.pack_weight(_dev(w, cuda), 1)
    dx0, dx1 = ops.conv3d_bwd_data(_cl(dy, cuda), wpb, (8, 8, 8), 32, 16, 3, 1, 1)
    assert _relerr(_np(dx0), dx_ref[:, :32]) < 1e-4
    assert _relerr(_np(dx1), dx_ref[:, 32:]) < 1e-4


WGRAD_CASES = [
    # N, C0, C1, size, Cout, k, s, p
    (1, 32, 0, (8, 8, 8), 32, 3, 1, 1),
    (2, 64, 0, (8, 8, 8), 64, 3, 1, 1),
    (1, 32, 32, (8, 8, 8), 32, 3, 1, 1),
    (1, 64, 0, (8, 8, 8), 32, 3, 1, 1),
    (1, 32, 0, (16, 16, 16), 32, 3, 2, 1),
    (1, 64, 0, (9, 9, 9), 64, 3, 2, 1),
    (2, 2, 0, (10, 9, 7), 32, 3, 1, 1),
    (1, 2, 0, (8, 8, 8), 2, 3, 1, 1),
    (1, 32, 0, (8, 8, 8), 1, 1, 1, 0),
    (1, 128, 0, (4, 4, 4), 72, 3, 1, 1),
    (1, 40, 0, (6, 6, 6), 24, 3, 1, 1),
]


@pytest.mark.parametrize("N,C0,C1,size,Cout,k,s,p", WGRAD_CASES)
def test_conv3d_bwd_weight_and_bias_match_oracle(cuda, N, C0, C1, size, Cout, k, s, p):
    rng = np.random.default_rng(21)
    Cin = C0 + C1
    x = rng.standard_normal((N, Cin, *size)).astype(np.float32)
    w = np.zeros((Cout, Cin, k, k, k), np.float32)
    osz = ops.conv_out_size(size, (k,) * 3, (s,) * 3, (p,) * 3)
    dy = rng.standard_normal((N, Cout, *osz)).astype(np.float32)
    _, dw_ref, db_ref = cops.conv3d_bwd(x, w, dy, s, p)
    x0 = _cl(x[:, :C0], cuda)
    x1 = _cl(x[:, C0:], cuda) if C1 else None
    dyd = _cl(dy, cuda)
    dw, db_fused = ops.conv3d_bwd_weight(x0, dyd, k, s, p, x1=x1, want_db=True)
    assert _relerr(_np(dw), dw_ref) < 2e-5
    assert _relerr(_np(db_fused), db_ref) < 2e-5
    db = ops.bias_grad(dyd)
    assert _relerr(_np(db), db_ref) < 2e-5


@pytest.mark.parametrize("Cin,Cout,size", [(32, 16, (4, 4, 4)), (64, 32, (8, 8, 8)), (8, 3, (5, 3, 2))])
def test_convtranspose_k2s2_bwd_weight(cuda, Cin, Cout, size):
    rng = np.random.default_rng(12)
    x = rng.standard_normal((2, Cin, *size)).astype(np.float32)
    w = np.zeros((Cin, Cout, 2, 2, 2), np.float32)
    dy = rng.standard_normal((2, Cout, *[2 * s for s in size])).astype(np.float32)
    _, dw_ref, _ = cops.conv_transpose3d_bwd(x, w, dy)
    dw = ops.convtranspose3d_k2s2_bwd_weight(_cl(x, cuda), _cl(dy, cuda))
    assert _relerr(_np(dw), dw_ref) < 2e-5


@pytest.mark.parametrize("Cin,Cout,size", [(32, 16, (4, 4, 4)), (64, 32, (8, 8, 8)), (8, 3, (5, 3, 2))])
def test_convtranspose_k2s2_fwd_and_bwd_data(cuda, Cin, Cout, size):
    rng = np.random.default_rng(11)
    x = rng.standard_normal((2, Cin, *size)).astype(np.float32)
    w = (rng.standard_normal((Cin, Cout, 2, 2, 2)) / np.sqrt(Cin)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    ref = cops.conv_transpose3d(x, w, b, 2, 0)
    wp = ops.pack_weight(_dev(w, cuda), 2)
    y = ops.convtranspose3d_k2s2_fwd(_cl(x, cuda), wp, _dev(b, cuda), Cout)
    assert _np(y).shape == ref.shape
    assert _relerr(_np(y), ref) < 1e-5
    dy = rng.standard_normal(ref.shape).astype(np.float32)
    dx_ref, _, _ = cops.conv_transpose3d_bwd(x, w, dy)
    wpb = ops.pack_weight(_dev(w, cuda), 3)
    dx = ops.convtranspose3d_k2s2_bwd_data(_cl(dy, cuda), wpb, Cin)
    assert _relerr(_np(dx), dx_ref) < 1e-5


@pytest.mark.parametrize("act", ["swish", "relu", "gelu", "identity", "sigmoid", "tanh"])
@pytest.mark.parametrize("C,size", [(32, (8, 8, 8)), (2, (9, 7, 5)), (3, (4, 4, 4))])
def test_instance_norm_act_fwd(cuda, act, C, size):
    rng = np.random.default_rng(3)
    x = (rng.standard_normal((2, C, *size)) * 3 + 1.5).astype(np.float32)
    ref = cops.norm_act(x, True, 1e-5, act)
    xd = _cl(x, cuda)
    mean, rstd = ops.instance_stats(xd, 1e-5)
    out = ops.norm_act_fwd(xd, mean, rstd, act)
    np.testing.assert_allclose(_np(out), ref, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("act", ["swish", "relu", "gelu", "identity"])
@pytest.mark.parametrize("C,size", [(32, (8, 8, 8)), (2, (9, 7, 5)), (64, (12, 12, 12))])
def test_instance_norm_act_bwd(cuda, act, C, size):
    rng = np.random.default_rng(4)
    x = (rng.standard_normal((2, C, *size)) * 2 + 0.5).astype(np.float32)
    g = rng.standard_normal(x.shape).astype(np.float32)
    ref = cops.norm_act_bwd(x, g, True, 1e-5, act)
    xd, gd = _cl(x, cuda), _cl(g, cuda)
    mean, rstd = ops.instance_stats(xd, 1e-5)
    dx, _, _ = ops.norm_act_bwd(xd, gd, mean, rstd, act)
    assert _relerr(_np(dx), ref) < 5e-5


def test_norm_act_bwd_with_dropout_matches_forward_mask(cuda):
    # d/dx of sum(out * g) by finite structure: with act=identity and no norm,
    # dx must equal g * mask / (1-p) where mask is the forward's mask.
    x = torch.ones((1, 32, 8, 8, 8), device=cuda)
    g = torch.randn((1, 32, 8, 8, 8), device=cuda)
    xd, gd = ops.ndhwc(x), ops.ndhwc(g)
    out = ops.norm_act_fwd(xd, None, None, "identity", drop_p=0.3, seed=9, rng_offset=1)
    dx, _, _ = ops.norm_act_bwd(xd, gd, None, None, "identity", drop_p=0.3, seed=9, rng_offset=1)
    assert torch.allclose(dx, gd * out, rtol=1e-6, atol=1e-6)


def test_dropout_mask_statistics_and_determinism(cuda):
    x = torch.ones((1, 32, 16, 16, 16), device=cuda)
    xd = ops.ndhwc(x)
    a = ops.norm_act_fwd(xd, None, None, "identity", drop_p=0.15, seed=42, rng_offset=3)
    b = ops.norm_act_fwd(xd, None, None, "identity", drop_p=0.15, seed=42, rng_offset=3)
    c = ops.norm_act_fwd(xd, None, None, "identity", drop_p=0.15, seed=42, rng_offset=4)
    assert torch.equal(a, b)
    assert not torch.equal(a, c)
    kept = (a != 0).float().mean().item()
    assert abs(kept - 0.85) < 0.01
    vals = torch.unique(a)
    assert torch.allclose(vals, torch.tensor([0.0, 1.0 / 0.85], device=cuda))


def _chi2_2x2(a, b):
    """Pearson chi-square (1 degree of freedom) of two boolean arrays' 2x2 contingency table."""
    a = a.reshape(-1).astype(np.float64)
    b = b.reshape(-1).astype(np.float64)
    n = a.size
    n11 = float((a * b).sum())
    n1_, n_1 = float(a.sum()), float(b.sum())
    tab = np.array([[n11, n1_ - n11], [n_1 - n11, n - n1_ - n_1 + n11]])
    exp = np.outer([n1_, n - n1_], [n_1, n - n_1]) / n
    return float(((tab - exp) ** 2 / exp).sum())


def test_dropout_per_channel_rates_and_independence(cuda):
    """The 7-round Philox mask (csrc/common.h): keep rate of every one of 64 channels, independence
    of x-neighbours, of neighbouring channels and of the masks of offsets o and o + 1 (2x2 contingency
    tables: chi-square with 1 degree of freedom, 99.99 % point = 15.1 -- a lost round of the
    generator shows up as a statistic in the thousands on 0.5 M pairs)."""
    p = 0.15
    x = ops.ndhwc(torch.ones((2, 64, 16, 16, 16), device=cuda))
    keep = []
    for off in (5, 6):
        out = ops.norm_act_fwd(x, None, None, "identity", drop_p=p, seed=1234, rng_offset=off)
        keep.append((_np(out) != 0))
    k0, k1 = keep
    nper = k0[:, 0].size
    rates = k0.mean(axis=(0, 2, 3, 4))
    sigma = np.sqrt(p * (1 - p) / nper)
    assert np.abs(rates - (1 - p)).max() < 5 * sigma, (rates.min(), rates.max(), sigma)
    assert abs(k0.mean() - (1 - p)) < 5 * np.sqrt(p * (1 - p) / k0.size)
    assert _chi2_2x2(k0[..., :-1], k0[..., 1:]) < 15.1          # neighbours along W
    assert _chi2_2x2(k0[:, :, :, :-1], k0[:, :, :, 1:]) < 15.1  # along H
    assert _chi2_2x2(k0[:, :-1], k0[:, 1:]) < 15.1              # neighbouring channels (same float4 or next)
    assert _chi2_2x2(k0[:, 0::4], k0[:, 3::4]) < 15.1           # first and last word of one Philox call
    assert _chi2_2x2(k0, k1) < 15.1                              # offsets o and o + 1
    assert _chi2_2x2(k0[0], k0[1]) < 15.1                        # batch items


@pytest.mark.parametrize("act", ["swish", "identity"])
@pytest.mark.parametrize("norm", [True, False])
def test_dropout_forward_backward_mask_identity_p015(cuda, act, norm):
    """Forward and backward regenerate the SAME mask at the configuration's p = 0.15 (ordering
    norm -> dropout -> activation, adn_fn.py:140-152): the mask is read off an identity-activation
    forward, the forward with the activation equals act(u * mask / 0.85), and the backward equals
    the p = 0 backward of a gradient masked by hand (through the activation's derivative at the
    dropped-out value, computed here with torch)."""
    g = torch.Generator(device=cuda).manual_seed(3)
    x = ops.ndhwc(torch.randn((2, 32, 12, 10, 9), device=cuda, generator=g) + 0.3)
    gy = ops.ndhwc(torch.randn((2, 32, 12, 10, 9), device=cuda, generator=g))
    mean = rstd = None
    if norm:
        mean, rstd = ops.instance_stats(x, 1e-5)
    kw = dict(seed=77, rng_offset=11)
    u = ops.norm_act_fwd(x, mean, rstd, "identity", drop_p=0.0, **kw)            # normalised values
    ud = ops.norm_act_fwd(x, mean, rstd, "identity", drop_p=0.15, **kw)
    keep = (ud != 0)
    assert abs(keep.float().mean().item() - 0.85) < 0.01
    m = keep.float() / 0.85
    assert torch.allclose(ud, u * m, rtol=1e-6, atol=1e-7)
    fn = torch.nn.functional.silu if act == "swish" else (lambda t: t)
    out = ops.norm_act_fwd(x, mean, rstd, act, drop_p=0.15, **kw)
    assert torch.allclose(out, fn(u * m), rtol=2e-6, atol=1e-6)
    # backward: dt = g * act'(u m) * m, then the normalisation's backward (p = 0, identity)
    um = (u * m).detach().requires_grad_(True)
    fn(um).backward(gy)
    dt = (um.grad * m).contiguous(memory_format=torch.channels_last_3d)
    dx, _, _ = ops.norm_act_bwd(x, gy, mean, rstd, act, drop_p=0.15, **kw)
    dx0, _, _ = ops.norm_act_bwd(x, dt, mean, rstd, "identity", drop_p=0.0, **kw)
    scale = float(dx0.abs().max())
    assert float((dx - dx0).abs().max()) <= 2e-5 * scale
    if not norm:
        assert torch.equal(dx == 0, ~keep | (dx0 == 0))


def test_expanded_view_is_refused(cuda):
    """ops._ptr, always on: a stride-0 dimension of size > 1 never reaches a kernel (the masked
    attention backward of round 4 read B*H*T*T floats from a T*T storage this way)."""
    x = torch.ones((1, 32, 4, 4, 4), device=cuda).expand(2, 32, 4, 4, 4)
    with pytest.raises(_lib.AdellHipError, match="stride-0"):
        ops._ptr(x)
    assert ops._ptr(torch.ones((1, 1, 4), device=cuda).expand(1, 1, 4)) is not None


def test_cpu_tensor_is_refused():
    with pytest.raises(_lib.AdellHipError):
        ops.norm_act_fwd(torch.zeros(1, 4, 2, 2, 2), None, None, "relu")


# ---- small-channel paths (csrc/conv_small.hip) ---------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("N,C0,C1,size,Cout,k,p", [
    (1, 2, 0, (20, 18, 33), 32, 3, 1), (2, 2, 0, (9, 7, 5), 2, 3, 1), (1, 1, 2, (8, 8, 17), 5, 3, 1),
    # rows of a multiple of four voxels: the four-voxels-per-thread kernel (2 -> 2, 2 -> 1)
    (2, 2, 0, (9, 7, 8), 2, 3, 1), (1, 2, 0, (5, 6, 4), 1, 3, 1), (2, 2, 0, (24, 20, 64), 2, 3, 1),
    (1, 4, 0, (6, 6, 6), 19, 3, 1), (1, 3, 0, (7, 5, 9), 8, 1, 0), (1, 2, 0, (8, 8, 8), 16, 3, 0)])
@pytest.mark.parametrize("prec", ["f16x3", "fp32"])
def test_small_cin_weight_gradient(cuda, prec, N, C0, C1, size, Cout, k, p):
    """Cin <= 4: the vector-ALU weight-gradient kernel behind both conv_bwd_weight entries."""
    rng = np.random.default_rng(C0 * 7 + Cout)
    Cin = C0 + C1
    x = rng.standard_normal((N, Cin, *size)).astype(np.float32)
    w = rng.standard_normal((Cout, Cin, k, k, k)).astype(np.float32)
    osz = [s + 2 * p - k + 1 for s in size]
    dy = rng.standard_normal((N, Cout, *osz)).astype(np.float32)
    _, dw_ref, db_ref = cops.conv3d_bwd(x, w, dy, 1, p)
    xd = torch.from_numpy(x).to(cuda)
    x0 = ops.ndhwc(xd[:, :C0].contiguous())
    x1 = ops.ndhwc(xd[:, C0:].contiguous()) if C1 else None
    dw, db = ops.conv3d_bwd_weight(x0, ops.ndhwc(torch.from_numpy(dy).to(cuda)), k, 1, p, x1=x1,
                                   want_db=True, f16x3=(prec == "f16x3"))
    assert np.abs(dw.cpu().numpy().reshape(dw_ref.shape) - dw_ref).max() < 2e-5 * np.abs(dw_ref).max()
    assert np.abs(db.cpu().numpy() - db_ref).max() < 2e-5 * np.abs(db_ref).max()


@pytest.mark.gpu
@pytest.mark.parametrize("N,C0,C1,size,Cout", [(1, 32, 0, (16, 16, 16), 1), (2, 8, 8, (5, 6, 7), 3),
                                               (1, 70, 0, (4, 4, 9), 4), (1, 3, 0, (8, 8, 8), 2)])
def test_conv1_small_head_fwd_bwd(cuda, N, C0, C1, size, Cout):
    """1x1x1 conv with Cout <= 4 (logits head) through HF.conv3d: forward, dX, dW, db."""
    from adell_mri_amd import functional as HF

    rng = np.random.default_rng(C0 + Cout)
    Cin = C0 + C1
    x = rng.standard_normal((N, Cin, *size)).astype(np.float32)
    w = rng.standard_normal((Cout, Cin, 1, 1, 1)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    ref = cops.conv3d(x, w, b, 1, 0)
    dy = rng.standard_normal(ref.shape).astype(np.float32)
    dx_ref, dw_ref, db_ref = cops.conv3d_bwd(x, w, dy, 1, 0)
    xd = torch.from_numpy(x).to(cuda)
    x0 = ops.ndhwc(xd[:, :C0].contiguous()).requires_grad_(True)
    x1 = ops.ndhwc(xd[:, C0:].contiguous()).requires_grad_(True) if C1 else None
    wd = torch.from_numpy(w).to(cuda).requires_grad_(True)
    bd = torch.from_numpy(b).to(cuda).requires_grad_(True)
    y = HF.conv3d(x0, wd, bd, 1, 0, x1=x1)
    assert not hasattr(y, "_adell_partials")
    y.backward(ops.ndhwc(torch.from_numpy(dy).to(cuda)))
    rel = lambda a, r: float(np.abs(a - r).max() / (np.abs(r).max() + 1e-30))  # noqa: E731
    assert rel(y.detach().cpu().numpy(), ref) < 1e-5
    dx = x0.grad.cpu().numpy() if x1 is None else np.concatenate(
        [x0.grad.cpu().numpy(), x1.grad.cpu().numpy()], 1)
    assert rel(dx, dx_ref) < 1e-5
    assert rel(wd.grad.cpu().numpy(), dw_ref) < 2e-5
    assert rel(bd.grad.cpu().numpy(), db_ref) < 2e-5


@pytest.mark.gpu
def test_residual_block_link_gradient_rides_the_head_conv(cuda):
    """functional.GradCarry: the gradient of `op(X) + X` (res_blocks.py:192) is added inside the
    backward-data epilogue of the block's first conv. Same gradients as the autograd-accumulated
    form, bit for bit in the weights, to fp32 rounding in dX (one add order differs)."""
    from adell_mri_amd import ops as _ops
    from adell_mri_amd.modules.layers.adn_fn import get_adn_fn
    from adell_mri_amd.modules.layers.res_blocks import ResidualBlock3d

    torch.manual_seed(0)
    out = {}
    for inter in (None, 16):
        blk = ResidualBlock3d(32, 3, inter, 32, get_adn_fn(3, "instance", "swish", 0.0)).to(cuda)
        x0 = torch.randn(2, 32, 16, 24, 16, device=cuda)
        r = torch.randn(2, 32, 16, 24, 16, device=cuda)
        for mode in ("carry", "autograd"):
            _ops.FLAGS["no_grad_carry"] = mode == "autograd"
            try:
                x = x0.clone().requires_grad_(True)
                blk.zero_grad()
                y = blk(x)
                (y * r).sum().backward()
            finally:
                _ops.FLAGS["no_grad_carry"] = False
            out[mode] = (y.detach(), x.grad.clone(),
                         {k: p.grad.clone() for k, p in blk.named_parameters()})
        assert torch.equal(out["carry"][0], out["autograd"][0])
        gx_c, gx_a = out["carry"][1], out["autograd"][1]
        assert float((gx_c - gx_a).abs().max()) <= 2e-6 * float(gx_a.abs().max())
        for k, g in out["autograd"][2].items():
            assert torch.equal(out["carry"][2][k], g), k
        # no gradient wanted for X: no carry, the block still runs
        with torch.no_grad():
            assert torch.equal(blk(x0), out["carry"][0])


@pytest.mark.gpu
@pytest.mark.parametrize("C,size,out", [(64, (10, 12, 14), (8, 9, 14)), (6, (7, 7, 9), (6, 5, 5)),
                                        (3, (5, 6, 7), (4, 6, 6)), (32, (9, 9, 9), (9, 9, 9))])
def test_crop3d_against_slicing(cuda, C, size, out):
    """functional.crop3d (adell_window_ndhwc both ways) against the strided view of crop_to_size
    (layers/utils.py:30-52): values and the zero-framed gradient, bit for bit."""
    from adell_mri_amd import functional as HF
    from adell_mri_amd.modules.layers.utils import crop_to_size

    g = torch.Generator().manual_seed(C)
    x = torch.randn((2, C, *size), generator=g)
    sl = [slice(None), slice(None)] + [slice((c - o) // 2, (c - o) // 2 + o) for c, o in zip(size, out)]
    xr = x.clone().requires_grad_(True)
    ref = xr[tuple(sl)]
    w = torch.randn(ref.shape, generator=g)
    (ref * w).sum().backward()
    xg = ops.ndhwc(x.to(cuda)).requires_grad_(True)
    got = crop_to_size(xg, list(out))
    if tuple(size) != tuple(out):
        assert got.permute(0, 2, 3, 4, 1).is_contiguous()
    (got * w.to(cuda)).sum().backward()
    assert torch.equal(got.detach().cpu(), ref.detach())
    assert torch.equal(xg.grad.cpu(), xr.grad)
    assert torch.equal(HF.crop3d(xg.detach(), out).cpu(), ref.detach())


@pytest.mark.gpu
@pytest.mark.parametrize("momentum", [0.1, None])
def test_batchnorm_running_statistics_against_torch(cuda, momentum):
    """Two training passes through functional.norm_drop_act(norm="batch") update running_mean /
    running_var / num_batches_tracked as torch.nn.BatchNorm3d does (adell_bn_running_update: one
    launch per site), with momentum 0.1 and with the cumulative average (momentum=None)."""
    from adell_mri_amd import functional as HF

    C = 6
    ref = torch.nn.BatchNorm3d(C, momentum=momentum)
    ref.train()
    rm, rv = torch.zeros(C, device=cuda), torch.ones(C, device=cuda)
    nbt = torch.zeros((), dtype=torch.int64, device=cuda)
    g = torch.Generator().manual_seed(3)
    for step in range(2):
        x = torch.randn((2, C, 5, 6, 7), generator=g) * (1.0 + step) + 0.5 * step
        want = ref(x)
        got = HF.norm_drop_act(ops.ndhwc(x.to(cuda)), norm="batch", eps=ref.eps, running=(rm, rv, nbt),
                               momentum=momentum, training=True)
        assert _relerr(_np(got), _np(want.detach())) < 2e-6
    assert int(nbt) == 2 and int(ref.num_batches_tracked) == 2
    assert _relerr(_np(rm), _np(ref.running_mean)) < 1e-6
    assert _relerr(_np(rv), _np(ref.running_var)) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,size,rate", [(8, 16, (10, 12, 8), 2), (4, 6, (9, 7, 11), 2),
                                                (16, 8, (9, 9, 10), 3), (8, 8, (7, 8, 9), (2, 1, 2))])
def test_dilated_conv_against_torch(cuda, cin, cout, size, rate):
    """functional.conv3d_dilated (space-to-batch gather + the plain conv kernels on the rate^3
    sub-lattices + inverse gather; zero frame for extents that are not multiples of the rate) against
    torch's dilated conv in fp64: values and the gradients of input, weight and bias."""
    from adell_mri_amd import functional as HF

    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn((2, cin, *size), generator=g, dtype=torch.float64).requires_grad_(True)
    w = (torch.randn((cout, cin, 3, 3, 3), generator=g, dtype=torch.float64) / np.sqrt(27 * cin)).requires_grad_(True)
    b = torch.randn(cout, generator=g, dtype=torch.float64).requires_grad_(True)
    ref = torch.nn.functional.conv3d(x, w, b, padding=rate, dilation=rate)
    dy = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    ref.backward(dy)
    xd = ops.ndhwc(x.detach().float().to(cuda)).requires_grad_(True)
    wd = w.detach().float().to(cuda).requires_grad_(True)
    bd = b.detach().float().to(cuda).requires_grad_(True)
    out = HF.conv3d_dilated(xd, wd, bd, rate)
    assert tuple(out.shape) == tuple(ref.shape)
    out.backward(dy.float().to(cuda))
    assert _relerr(_np(out).astype(np.float64), ref.detach().numpy()) < 5e-6
    assert _relerr(_np(xd.grad).astype(np.float64), x.grad.numpy()) < 5e-6
    assert _relerr(_np(wd.grad).astype(np.float64), w.grad.numpy()) < 2e-5
    assert _relerr(_np(bd.grad).astype(np.float64), b.grad.numpy()) < 2e-5
