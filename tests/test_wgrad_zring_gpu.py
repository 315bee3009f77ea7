"""The z-marching backward-weight kernel (conv_wgrad_zring.hip) against the per-plane f16x3 kernel
and the fp32-MFMA kernel: ragged planes, batch > 1, virtual concat, padding 0 and 1."""
import ctypes
import os

import pytest
import torch

from adell_mri_amd import _lib, ops

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


@pytest.mark.parametrize("n,c0,c1,cout,size,pad", [(1, 32, 0, 32, (16, 24, 40), 1),
                                                    (2, 32, 0, 64, (12, 20, 28), 1),
                                                    (1, 32, 32, 32, (9, 17, 33), 1),
                                                    (1, 64, 0, 64, (24, 16, 16), 1),
                                                    (1, 32, 0, 32, (14, 18, 22), 0),
                                                    (1, 32, 0, 32, (64, 64, 64), 1),
                                                    # channel counts in whole 16s: ragged last
                                                    # tile, tile straddling the concat boundary
                                                    (1, 16, 0, 16, (16, 24, 40), 1),
                                                    (1, 16, 0, 32, (12, 20, 28), 1),
                                                    (2, 32, 0, 16, (12, 20, 28), 1),
                                                    (1, 16, 16, 32, (9, 17, 33), 1),
                                                    (1, 48, 0, 48, (14, 18, 22), 0),
                                                    (2, 16, 32, 16, (10, 16, 24), 1),
                                                    (1, 48, 16, 80, (10, 16, 24), 1),
                                                    (1, 16, 0, 16, (48, 48, 48), 1)])
def test_zring_matches_plane_kernel_and_fp32(cuda, n, c0, c1, cout, size, pad):
    g = torch.Generator().manual_seed(c0 + cout + size[0])
    D, H, W = size
    x0 = ops.ndhwc((torch.randn(n, c0, D, H, W, generator=g) * 2).to(cuda))
    x1 = ops.ndhwc(torch.randn(n, c1, D, H, W, generator=g).to(cuda)) if c1 else None
    Do, Ho, Wo = (s + 2 * pad - 2 for s in size)
    dy = ops.ndhwc((torch.randn(n, cout, Do, Ho, Wo, generator=g) * 1e-3).to(cuda))

    plan = (ctypes.c_int * 8)()      # WgradZrPlan: ntx, nty, nseg, seglen, nci, nco, R, t16
    assert _lib.lib().adell_wgrad_zring_plan(n, D, H, W, c0, c1, cout, 3, 3, 3, 1, 1, 1, Do, Ho, Wo,
                                             plan) == 1
    # layers with 16 channels on either side run the 16 x 16 tile form (v_mfma_f32_16x16x32_f16)
    assert plan[7] == (1 if (c0 + c1 == 16 or cout == 16) else 0)

    def run(f16):
        return ops.conv3d_bwd_weight(x0, dy, 3, 1, pad, x1=x1, want_db=True, f16x3=f16)

    dw_z, db_z = run(True)
    with _lib.tuning(wgrad_nozring=1):
        dw_p, db_p = run(True)
    dw_32, db_32 = run(False)
    assert _rel(dw_z, dw_32) < 2e-5
    assert _rel(dw_z, dw_p) < 2e-5
    assert _rel(db_z, db_32) < 2e-5
    # deterministic: same slabs, same fold order
    dw_z2, db_z2 = run(True)
    assert torch.equal(dw_z, dw_z2) and torch.equal(db_z, db_z2)
    if plan[7]:
        # the 32 x 32 tile form of the same march (what these layers ran on before)
        with _lib.tuning(wgrad_no16=1):
            dw_w, db_w = run(True)
        assert _rel(dw_z, dw_w) < 2e-5 and _rel(db_z, db_w) < 2e-5


@pytest.mark.parametrize("n,c0,c1,cout,size", [(2, 16, 0, 16, (16, 24, 40)), (1, 32, 0, 16, (12, 20, 28)),
                                               (1, 16, 0, 32, (12, 16, 24)), (2, 16, 16, 16, (10, 16, 24))])
def test_zring16_against_fp64(cuda, n, c0, c1, cout, size):
    """The 16 x 16 tile form against torch's fp64 weight gradient on the CPU (operands that are
    NOT well scaled: the power-of-two block scaling has to carry them)."""
    g = torch.Generator().manual_seed(7 * c0 + cout + size[1])
    D, H, W = size
    x = torch.randn(n, c0 + c1, D, H, W, generator=g) * 37.0
    dy = torch.randn(n, cout, D, H, W, generator=g) * 3e-4
    w = torch.zeros(cout, c0 + c1, 3, 3, 3, dtype=torch.float64, requires_grad=True)
    b = torch.zeros(cout, dtype=torch.float64, requires_grad=True)
    y = torch.nn.functional.conv3d(x.double(), w, b, padding=1)
    y.backward(dy.double())
    x0 = ops.ndhwc(x[:, :c0].contiguous().to(cuda))
    x1 = ops.ndhwc(x[:, c0:].contiguous().to(cuda)) if c1 else None
    dw, db = ops.conv3d_bwd_weight(x0, ops.ndhwc(dy.to(cuda)), 3, 1, 1, x1=x1, want_db=True, f16x3=True)
    assert _rel(dw.cpu().double(), w.grad) < 3e-6
    assert _rel(db.cpu().double(), b.grad) < 3e-6
