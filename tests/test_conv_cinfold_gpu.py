"""The K = 27 Cin im2col-GEMM kernels of the narrow-input convs (csrc/conv_cinfold.hip) against the
C oracle (fp64 accumulation): forward with bias and the statistics partials, weight and bias
gradients; every Cin in 1..4, ragged bricks, padding 0 / 1, batch > 1, output widths on both sides
of the 32-column tile; and the layer dispatch (functional.conv3d takes this path, the x-tap fold
stays available behind FLAGS)."""
import numpy as np
import pytest
import torch

from adell_mri_amd import functional as HF
from adell_mri_amd import ops
from oracle import cops

pytestmark = pytest.mark.gpu

CASES = [  # N, Cin, size, Cout, pad
    (2, 2, (16, 16, 16), 32, 1), (1, 2, (40, 12, 9), 64, 1), (1, 1, (33, 8, 8), 16, 0), (1, 1, (9, 13, 11), 16, 1), (1, 3, (8, 8, 8), 40, 1),
    (1, 4, (12, 10, 9), 72, 1), (2, 2, (10, 11, 12), 32, 0), (1, 2, (5, 6, 7), 8, 1),
    (1, 1, (32, 32, 32), 16, 1), (3, 2, (4, 4, 4), 33, 1)]


def _rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


@pytest.mark.parametrize("N,Cin,size,Cout,pad", CASES)
def test_cinfold_forward_and_weight_gradient_match_oracle(cuda, N, Cin, size, Cout, pad):
    rng = np.random.default_rng(Cin * 100 + Cout)
    x = rng.standard_normal((N, Cin, *size)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, 3, 3, 3)) / np.sqrt(27 * Cin)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    ref = cops.conv3d(x, w, b, 1, pad)
    dy = rng.standard_normal(ref.shape).astype(np.float32)
    dx_ref, dw_ref, db_ref = cops.conv3d_bwd(x, w, dy, 1, pad)
    xd = ops.ndhwc(torch.from_numpy(x).to(cuda))
    wd, bd = torch.from_numpy(w).to(cuda), torch.from_numpy(b).to(cuda)
    assert ops.conv_cinfold_ok(wd, xd, None, (1, 1, 1), (pad,) * 3, None) == (Cout > 4)
    y, part = ops.conv_cinfold_fwd(xd, wd, bd, (pad,) * 3, True)
    assert _rel(y.cpu().numpy(), ref) < 2e-6
    # the split-f16 MFMA form (two input channels; the others run the same exact kernel): ~2^-22 per
    # product, and statistics partials of what it wrote
    y16, part16 = ops.conv_cinfold_fwd(xd, wd, bd, (pad,) * 3, True, f16x3=True)
    assert _rel(y16.cpu().numpy(), ref) < 3e-6
    m16, r16 = ops.stats_finalize(part16, int(np.prod(ref.shape[2:])), 1e-5)
    np.testing.assert_allclose(m16.cpu().numpy(), ref.reshape(N, Cout, -1).mean(-1), rtol=1e-3,
                               atol=1e-4)
    if Cin == 2:    # a large dynamic range inside one tensor: the per-brick scale keeps 22 bits
        big = xd * torch.logspace(-6, 6, xd.shape[2], device=cuda).view(1, 1, -1, 1, 1)
        refb = cops.conv3d(big.cpu().numpy(), w, b, 1, pad)
        yb, _ = ops.conv_cinfold_fwd(ops.ndhwc(big), wd, bd, (pad,) * 3, False, f16x3=True)
        plane = np.abs(refb).max(axis=(0, 1, 3, 4), keepdims=True) + 1e-30
        assert float((np.abs(yb.cpu().numpy() - refb) / plane).max()) < 2e-5
    V = int(np.prod(ref.shape[2:]))
    mean, rstd = ops.stats_finalize(part, V, 1e-5)
    np.testing.assert_allclose(mean.cpu().numpy(), ref.reshape(N, Cout, -1).mean(-1), rtol=1e-3,
                               atol=1e-4)
    np.testing.assert_allclose(rstd.cpu().numpy(),
                               1 / np.sqrt(ref.reshape(N, Cout, -1).var(-1) + 1e-5), rtol=1e-3)
    dyd = ops.ndhwc(torch.from_numpy(dy).to(cuda))
    dw, db = ops.conv_cinfold_bwd_weight(xd, dyd, (pad,) * 3, True)
    assert _rel(dw.cpu().numpy(), dw_ref) < 5e-6
    assert _rel(db.cpu().numpy(), db_ref) < 5e-6
    dx = ops.conv_cinfold_bwd_data(dyd, wd, size, (pad,) * 3)
    dx16 = ops.conv_cinfold_bwd_data(dyd, wd, size, (pad,) * 3, f16x3=True)   # split-f16 GEMM
    if Cout <= 64 and Cout % 4 == 0:
        assert _rel(dx.cpu().numpy(), dx_ref) < 5e-6
        assert _rel(dx16.cpu().numpy(), dx_ref) < 5e-6
    else:
        assert dx is None and dx16 is None
    dw2, none = ops.conv_cinfold_bwd_weight(xd, dyd, (pad,) * 3, False)
    assert none is None and torch.equal(dw, dw2)      # deterministic fold order
    # the split-f16 MFMA weight gradient (every channel count): ~2^-22 per product, the bias gradient
    # is the same fp32 sum; badly scaled operands (the per-slice / per-tensor scales carry them)
    dw16, db16 = ops.conv_cinfold_bwd_weight(xd, dyd, (pad,) * 3, True, f16x3=True)
    assert _rel(dw16.cpu().numpy(), dw_ref) < 5e-6
    assert _rel(db16.cpu().numpy(), db_ref) < 5e-6
    dw16b, _ = ops.conv_cinfold_bwd_weight(xd, dyd, (pad,) * 3, False, f16x3=True)
    assert torch.equal(dw16, dw16b)
    dws, _ = ops.conv_cinfold_bwd_weight(xd * 3e4, dyd * 2e-5, (pad,) * 3, False, f16x3=True)
    assert _rel(dws.cpu().numpy() / (3e4 * 2e-5), dw_ref) < 5e-6


def test_layer_takes_the_cinfold_path_and_matches_the_fold_path(cuda):
    """functional.conv3d on a 2 -> 32 conv: same results (to fp32 rounding) through the im2col-GEMM
    kernels as through the x-tap fold + f16x3 kernels they replace."""
    g = torch.Generator().manual_seed(5)
    x = ops.ndhwc(torch.randn(2, 2, 24, 20, 28, generator=g).to(cuda))
    w0 = (torch.randn(32, 2, 3, 3, 3, generator=g) * 0.2).to(cuda)
    b0 = torch.randn(32, generator=g).to(cuda)
    r = ops.ndhwc(torch.randn(2, 32, 24, 20, 28, generator=g).to(cuda))
    out = {}
    for mode in ("cinfold", "fold"):
        ops.FLAGS["no_cinfold"] = mode == "fold"
        try:
            w, b = w0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
            ops.KERNEL_TIMER = ops.KernelTimer()
            y = HF.conv3d(x, w, b, 1, 1)
            (y * r).sum().backward()
            names = set(ops.KERNEL_TIMER.summary())
        finally:
            ops.FLAGS["no_cinfold"] = False
            ops.KERNEL_TIMER = None
        assert ("adell_cinfold_kernel" in names) == (mode == "cinfold")
        out[mode] = (y.detach(), w.grad, b.grad)
    for a, bb in zip(out["cinfold"], out["fold"]):
        assert float((a - bb).abs().max()) <= 2e-5 * float(bb.abs().max())
