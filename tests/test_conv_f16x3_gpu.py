"""The f16x3 (error-compensated f16 MFMA) conv path against the fp64-accumulated C
oracle: fp32-class accuracy (tolerance 5e-6 relative to the output scale) including
badly scaled operands, which exercise the in-kernel power-of-two scaling."""
import numpy as np
import pytest
import torch

from adell_mri_amd import _lib, ops
from oracle import cops

pytestmark = pytest.mark.gpu


def _dev(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def _cl(a, device):
    return ops.ndhwc(_dev(a, device))


def _np(t):
    return t.detach().contiguous().cpu().numpy()


def _relerr(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


CASES = [
    # N, Cin, size, Cout, k, s, p, cfg, xscale, wscale
    (1, 32, (16, 16, 16), 32, 3, 1, 1, -1, 1.0, 1.0),
    (2, 2, (12, 10, 9), 32, 3, 1, 1, -1, 1.0, 1.0),
    (1, 64, (8, 8, 8), 64, 3, 1, 1, -1, 1.0, 1.0),
    (1, 32, (16, 16, 16), 32, 3, 2, 1, -1, 1.0, 1.0),
    (1, 16, (17, 15, 13), 48, 3, 2, 1, -1, 1.0, 1.0),
    (1, 32, (8, 8, 8), 1, 1, 1, 0, -1, 1.0, 1.0),
    (1, 40, (8, 8, 8), 72, 3, 1, 1, -1, 1.0, 1.0),
    (1, 32, (16, 16, 16), 64, 3, 1, 1, 0, 1.0, 1.0),
    (1, 32, (16, 16, 16), 32, 3, 1, 1, 1, 1.0, 1.0),
    (1, 32, (16, 16, 16), 64, 3, 1, 1, 2, 1.0, 1.0),
    (1, 32, (16, 16, 16), 32, 3, 1, 1, 3, 1.0, 1.0),
    (1, 32, (8, 8, 8), 32, 3, 1, 1, -1, 1e-9, 1e3),     # gradient-sized inputs
    (1, 32, (8, 8, 8), 32, 3, 1, 1, -1, 3e7, 1e-6),     # huge activations, tiny weights
]


@pytest.mark.parametrize("N,Cin,size,Cout,k,s,p,cfg,xs,ws", CASES)
def test_conv3d_fwd_f16x3(cuda, N, Cin, size, Cout, k, s, p, cfg, xs, ws):
    rng = np.random.default_rng(1234)
    x = (rng.standard_normal((N, Cin, *size)) * xs).astype(np.float32)
    # channels with very different magnitudes inside one 16-channel chunk
    x[:, ::3] *= 1e-3
    w = (rng.standard_normal((Cout, Cin, k, k, k)) * ws / np.sqrt(Cin * k ** 3)).astype(np.float32)
    b = (rng.standard_normal(Cout) * xs * ws).astype(np.float32)
    ref = cops.conv3d(x, w, b, s, p)
    _lib.lib().adell_debug_force_conv_cfg(cfg)
    try:
        wp = ops.pack_weight_f16x3(_dev(w, cuda), 0)
        y, part = ops.conv3d_fwd(_cl(x, cuda), wp, _dev(b, cuda), Cout, k, s, p, want_stats=True)
        torch.cuda.synchronize()
    finally:
        _lib.lib().adell_debug_force_conv_cfg(-1)
    assert _relerr(_np(y), ref) < 5e-6
    V = np.prod(ref.shape[2:])
    mean, rstd = ops.stats_finalize(part, V, 1e-5 * (xs * ws) ** 2)
    np.testing.assert_allclose(_np(mean), ref.reshape(N, Cout, -1).mean(-1), rtol=1e-3,
                               atol=1e-4 * xs * ws)


@pytest.mark.gpu
def test_wide_bottom_level_plan_against_torch(cuda):
    """512 input channels on 2 048 .. 4 095 voxels (the 9 x 9 x 33 bottom level of the ResNet-backbone
    U-Net, csrc/conv3d.hip adell_plan_f16: 256-voxel x 64-column bricks instead of the two-wave
    64 x 32 ones): forward with statistics and backward-data against torch's fp64 convolution."""
    g = torch.Generator().manual_seed(5)
    N, Cin, size, Cout = 1, 512, (9, 9, 26), 64
    x = torch.randn((N, Cin, *size), generator=g)
    w = torch.randn((Cout, Cin, 3, 3, 3), generator=g) / np.sqrt(Cin * 27)
    b = torch.randn(Cout, generator=g)
    ref = torch.nn.functional.conv3d(x.double(), w.double(), b.double(), 1, 1)
    wp = ops.pack_weight_f16x3(w.to(cuda), 0)
    y, part = ops.conv3d_fwd(ops.ndhwc(x.to(cuda)), wp, b.to(cuda), Cout, 3, 1, 1, want_stats=True)
    assert _relerr(_np(y), ref.detach().numpy()) < 5e-6
    mean, _ = ops.stats_finalize(part, int(np.prod(size)), 1e-5)
    np.testing.assert_allclose(_np(mean), ref.detach().reshape(N, Cout, -1).mean(-1).numpy(), rtol=1e-3,
                               atol=1e-4)
    # backward-data of a 64 -> 512 conv: the gradient operand has the 512 channels
    w2 = torch.randn((Cin, Cout, 3, 3, 3), generator=g) / np.sqrt(Cin * 27)
    dy = torch.randn((N, Cin, *size), generator=g)
    xd = torch.zeros((N, Cout, *size), dtype=torch.float64, requires_grad=True)
    torch.nn.functional.conv3d(xd, w2.double(), None, 1, 1).backward(dy.double())
    wpb = ops.pack_weight_f16x3(w2.to(cuda), 1)
    dx, _ = ops.conv3d_bwd_data(ops.ndhwc(dy.to(cuda)), wpb, size, Cout, 0, 3, 1, 1)
    assert _relerr(_np(dx), xd.grad.numpy()) < 5e-6


@pytest.mark.parametrize("N,Cin,size,Cout,k,s,p,gs", [
    (1, 32, (8, 8, 8), 32, 3, 1, 1, 1.0), (1, 16, (16, 16, 16), 24, 3, 2, 1, 1e-8),
    (1, 8, (9, 9, 9), 8, 3, 2, 1, 1.0), (2, 2, (8, 8, 8), 32, 3, 1, 1, 1e-7)])
def test_conv3d_bwd_data_f16x3(cuda, N, Cin, size, Cout, k, s, p, gs):
    rng = np.random.default_rng(99)
    x = rng.standard_normal((N, Cin, *size)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, k, k, k)) / np.sqrt(Cout * k ** 3)).astype(np.float32)
    osz = ops.conv_out_size(size, (k,) * 3, (s,) * 3, (p,) * 3)
    dy = (rng.standard_normal((N, Cout, *osz)) * gs).astype(np.float32)
    dx_ref, _, _ = cops.conv3d_bwd(x, w, dy, s, p)
    wpb = ops.pack_weight_f16x3(_dev(w, cuda), 1)
    dx0, _ = ops.conv3d_bwd_data(_cl(dy, cuda), wpb, size, Cin, 0, k, s, p)
    assert _relerr(_np(dx0), dx_ref) < 5e-6


def test_concat_residual_f16x3(cuda):
    rng = np.random.default_rng(7)
    xa = rng.standard_normal((1, 32, 8, 8, 8)).astype(np.float32)
    xb = rng.standard_normal((1, 32, 8, 8, 8)).astype(np.float32)
    w = (rng.standard_normal((64, 64, 3, 3, 3)) * 0.03).astype(np.float32)
    b = rng.standard_normal(64).astype(np.float32)
    res = rng.standard_normal((1, 64, 8, 8, 8)).astype(np.float32)
    ref = cops.conv3d(np.concatenate([xa, xb], 1), w, b, 1, 1) + res
    wp = ops.pack_weight_f16x3(_dev(w, cuda), 0)
    y, _ = ops.conv3d_fwd(_cl(xa, cuda), wp, _dev(b, cuda), 64, 3, 1, 1, x1=_cl(xb, cuda),
                          residual=_cl(res, cuda))
    assert _relerr(_np(y), ref) < 5e-6


WGRAD_CASES = [
    # N, C0, C1, size, Cout, k, s, p, gscale
    (1, 32, 0, (8, 8, 8), 32, 3, 1, 1, 1.0),
    (2, 64, 0, (8, 8, 8), 64, 3, 1, 1, 1e-7),
    (1, 32, 32, (8, 8, 8), 32, 3, 1, 1, 1.0),
    (1, 64, 0, (8, 8, 8), 32, 3, 1, 1, 1.0),
    (1, 32, 0, (16, 16, 16), 32, 3, 2, 1, 1.0),
    (1, 64, 0, (9, 9, 9), 64, 3, 2, 1, 1e-6),
    (2, 2, 0, (10, 9, 7), 32, 3, 1, 1, 1.0),
    (1, 2, 0, (8, 8, 8), 2, 3, 1, 1, 1.0),
    (1, 32, 0, (8, 8, 8), 1, 1, 1, 0, 1.0),
    (1, 128, 0, (4, 4, 4), 72, 3, 1, 1, 1.0),
    (1, 40, 0, (6, 6, 6), 24, 3, 1, 1, 1.0),
    (1, 16, 0, (5, 1, 9), 16, 3, 1, 1, 1.0),
]


@pytest.mark.parametrize("N,C0,C1,size,Cout,k,s,p,gs", WGRAD_CASES)
def test_conv3d_bwd_weight_f16x3(cuda, N, C0, C1, size, Cout, k, s, p, gs):
    rng = np.random.default_rng(21)
    Cin = C0 + C1
    x = (rng.standard_normal((N, Cin, *size)) * 3).astype(np.float32)
    w = np.zeros((Cout, Cin, k, k, k), np.float32)
    osz = ops.conv_out_size(size, (k,) * 3, (s,) * 3, (p,) * 3)
    dy = (rng.standard_normal((N, Cout, *osz)) * gs).astype(np.float32)
    _, dw_ref, db_ref = cops.conv3d_bwd(x, w, dy, s, p)
    x0 = _cl(x[:, :C0], cuda)
    x1 = _cl(x[:, C0:], cuda) if C1 else None
    dw, db = ops.conv3d_bwd_weight(x0, _cl(dy, cuda), k, s, p, x1=x1, want_db=True, f16x3=True)
    assert _relerr(_np(dw), dw_ref) < 5e-6
    assert _relerr(_np(db), db_ref) < 2e-5
