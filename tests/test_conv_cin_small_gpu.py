"""Small-Cin (<= 4) convolution kernels (csrc/conv_small.hip) against torch CPU fp64."""
import pytest
import torch
import torch.nn.functional as F

from adell_mri_amd import functional as HF
from adell_mri_amd import ops

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


@pytest.mark.parametrize("n,cin,cout,size,kd,pad,bias", [(1, 2, 2, (9, 17, 21), 3, 1, True),
                                                         # rows of a multiple of 4 voxels: the
                                                         # four-voxels-per-thread forward (2 -> 2, 2 -> 1)
                                                         (2, 2, 2, (9, 17, 20), 3, 1, True),
                                                         (1, 2, 1, (6, 5, 8), 3, 1, False),
                                                         (1, 2, 2, (5, 7, 4), 3, 1, True),
                                                         (1, 2, 2, (40, 36, 64), 3, 1, True),
                                                         (2, 2, 32, (8, 12, 19), 3, 1, True),
                                                         (1, 1, 16, (1, 33, 30), 1, (0, 1, 1), True),
                                                         (1, 3, 40, (7, 9, 11), 3, 0, False),
                                                         (1, 4, 8, (6, 10, 34), 3, 1, True),
                                                         # Cin, Cout <= 2: voxel-per-thread dW kernel
                                                         (2, 1, 2, (5, 9, 70), 3, 1, True),
                                                         (1, 2, 1, (6, 7, 9), 3, 0, True),
                                                         (3, 1, 1, (4, 5, 6), 3, 1, False),
                                                         (2, 2, 2, (16, 24, 40), 3, 1, True)])
def test_cin_small_fwd_stats_and_grads_match_torch(cuda, monkeypatch, n, cin, cout, size, kd, pad,
                                                   bias):
    monkeypatch.setitem(ops.FLAGS, "cin_small_all", True)   # every width, not only Cout <= 4
    g = torch.Generator().manual_seed(cin * 10 + cout)
    x = torch.randn(n, cin, *size, generator=g, dtype=torch.float64).requires_grad_(True)
    w = (torch.randn(cout, cin, kd, 3, 3, generator=g, dtype=torch.float64) * 0.2).requires_grad_(True)
    b = torch.randn(cout, generator=g, dtype=torch.float64).requires_grad_(True) if bias else None
    y = F.conv3d(x, w, b, padding=pad)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(dy)
    xd = ops.ndhwc(x.detach().float().to(cuda)).requires_grad_(True)
    wd = w.detach().float().to(cuda).requires_grad_(True)
    bd = b.detach().float().to(cuda).requires_grad_(True) if bias else None
    assert ops.conv_cin_small_ok(wd, xd, None, (1, 1, 1), ops._triple(pad), None)
    yd = HF.conv3d(xd, wd, bd, stride=1, padding=pad, want_stats=True)
    assert _rel(yd.detach().cpu().double(), y.detach()) < 1e-5
    part = yd._adell_partials.double().sum(1).cpu()     # [N, Cout, 2]
    want = torch.stack([y.detach().sum((2, 3, 4)), (y.detach() ** 2).sum((2, 3, 4))], -1)
    assert _rel(part, want) < 1e-5
    yd.backward(ops.ndhwc(dy.float().to(cuda)))
    assert _rel(wd.grad.cpu().double(), w.grad) < 2e-5
    if bias:
        assert _rel(bd.grad.cpu().double(), b.grad) < 2e-5
    if cout % 4 == 0:
        assert _rel(xd.grad.cpu().double(), x.grad) < 1e-5


@pytest.mark.parametrize("n,cin,cout,size,k,pad", [(1, 2, 32, (12, 14, 37), 3, 1),
                                                   (2, 2, 64, (10, 9, 21), 7, 3),
                                                   (1, 1, 16, (6, 8, 19), 5, 2),
                                                   (1, 4, 40, (7, 9, 11), 3, 0),
                                                   (1, 3, 32, (9, 9, 9), (3, 3, 5), (1, 1, 2))])
def test_folded_x_taps_forward_matches_torch_and_plain_path(cuda, monkeypatch, n, cin, cout, size,
                                                            k, pad):
    """Kw * Cin <= 16: the forward folds the x taps into the 16-channel MFMA chunk
    (ops.fold_x_taps + a Kd x Kh x 1 conv); same output, statistics and gradients as the plain
    path and as torch."""
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn(n, cin, *size, generator=g, dtype=torch.float64).requires_grad_(True)
    kk = (k,) * 3 if isinstance(k, int) else k
    w = (torch.randn(cout, cin, *kk, generator=g, dtype=torch.float64) * 0.1).requires_grad_(True)
    b = torch.randn(cout, generator=g, dtype=torch.float64).requires_grad_(True)
    y = F.conv3d(x, w, b, padding=pad)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(dy)

    def run():
        xd = ops.ndhwc(x.detach().float().to(cuda)).requires_grad_(True)
        wd = w.detach().float().to(cuda).requires_grad_(True)
        bd = b.detach().float().to(cuda).requires_grad_(True)
        yd = HF.conv3d(xd, wd, bd, stride=1, padding=pad, want_stats=True)
        part = yd._adell_partials.double().sum(1).cpu()
        yd.backward(ops.ndhwc(dy.float().to(cuda)))
        return yd.detach().cpu().double(), part, xd.grad.cpu().double(), wd.grad.cpu().double()

    yf, pf, dxf, dwf = run()
    monkeypatch.setitem(HF.FLAGS, "no_fold", True)
    yp, pp, dxp, dwp = run()
    assert _rel(yf, y.detach()) < 5e-6 and _rel(yf, yp) < 5e-6
    want = torch.stack([y.detach().sum((2, 3, 4)), (y.detach() ** 2).sum((2, 3, 4))], -1)
    assert _rel(pf, want) < 1e-5
    assert _rel(dxf, x.grad) < 2e-5 and _rel(dwf, w.grad) < 2e-5
    assert torch.equal(dxf, dxp) and torch.equal(dwf, dwp)   # the backward is the same code
