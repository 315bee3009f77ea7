"""Backward-weight of the U-Net downsampling layer (32 -> 32 channels, k = 3, stride 2, padding 1;
unet.py:571-579) on its sub-lattice-walk kernel (csrc/conv_wgrad_s2.hip) against torch's fp64 conv
backward and the generic weight-gradient kernel, on whole and ragged bricks, with the bias gradient,
with and without caller-provided absmax words."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)


@pytest.mark.parametrize("n,size", [(2, (16, 16, 16)), (1, (12, 20, 10)), (1, (8, 8, 72)),
                                    (3, (2, 2, 2)), (1, (64, 64, 64)), (5, (8, 16, 8))])
@pytest.mark.parametrize("words", [False, True])
def test_wgrad_matches_fp64_and_the_generic_kernel(cuda, n, size, words):
    """(the planner takes this kernel from 8 bricks per CU upwards; whichever kernel runs, the
    result is checked against fp64 -- the kernel itself is exercised by
    see test_large_problem_takes_the_sublattice_kernel)"""
    from adell_mri_amd import _lib, ops

    g = torch.Generator().manual_seed(size[2] + n)
    x = torch.randn(n, 32, *size, generator=g)
    osz = ops.conv_out_size(size, (3,) * 3, (2,) * 3, (1,) * 3)
    dy = torch.randn(n, 32, *osz, generator=g) * 1e-3
    xr = x.double()
    wr = torch.zeros(32, 32, 3, 3, 3, dtype=torch.float64, requires_grad=True)
    br = torch.zeros(32, dtype=torch.float64, requires_grad=True)
    torch.nn.functional.conv3d(xr, wr, br, stride=2, padding=1).backward(dy.double())
    xd, dyd = ops.ndhwc(x.to(cuda)), ops.ndhwc(dy.to(cuda))
    xa = ops.absmax_word(xd) if words else None
    ya = ops.absmax_word(dyd) if words else None
    dw, db = ops.conv3d_bwd_weight(xd, dyd, 3, 2, 1, want_db=True, f16x3=True, x_amax=xa, dy_amax=ya)
    assert _rel(dw.view(32, 32, 3, 3, 3).cpu().double(), wr.grad) < 5e-6
    assert _rel(db.cpu().double(), br.grad) < 5e-6
    # the generic kernel (the z-ring A/B switch also turns this kernel off)
    L = _lib.lib()
    L.adell_set_tuning(b"wgrad_nozring", 1)
    try:
        dw2, db2 = ops.conv3d_bwd_weight(xd, dyd, 3, 2, 1, want_db=True, f16x3=True)
    finally:
        L.adell_set_tuning(b"wgrad_nozring", 0)
    assert _rel(dw, dw2) < 5e-6 and _rel(db, db2) < 5e-6
    # deterministic
    dw3, _ = ops.conv3d_bwd_weight(xd, dyd, 3, 2, 1, want_db=True, f16x3=True, x_amax=xa, dy_amax=ya)
    assert torch.equal(dw, dw3)


def test_other_shapes_keep_their_kernels(cuda):
    """64 channels, padding 0, stride 1: not this kernel (results against fp64 all the same)."""
    from adell_mri_amd import ops

    g = torch.Generator().manual_seed(1)
    for cin, cout, stride, pad in ((64, 32, 2, 1), (32, 32, 2, 0), (32, 32, 1, 1)):
        x = torch.randn(1, cin, 12, 12, 12, generator=g)
        wr = torch.zeros(cout, cin, 3, 3, 3, dtype=torch.float64, requires_grad=True)
        y = torch.nn.functional.conv3d(x.double(), wr, None, stride=stride, padding=pad)
        dy = torch.randn(y.shape, generator=g)
        y.backward(dy.double())
        dw = ops.conv3d_bwd_weight(ops.ndhwc(x.to(cuda)), ops.ndhwc(dy.to(cuda)), 3, stride, pad,
                                   f16x3=True)
        assert _rel(dw.view(wr.shape).cpu().double(), wr.grad) < 5e-6


def test_large_problem_takes_the_sublattice_kernel(cuda):
    """2 x 64 x 128 x 128 input: 2048 bricks of 8 x 8 x 2 -> csrc/conv_wgrad_s2.hip (checked through the
    workspace size, which follows the larger of the plans), against fp64."""
    import ctypes

    from adell_mri_amd import _lib, ops

    d_small = ops.make_conv_desc(2, (64, 64, 64), 32, 0, 32, 3, 2, 1)
    d_big = ops.make_conv_desc(2, (64, 128, 128), 32, 0, 32, 3, 2, 1)
    L = _lib.lib()
    ws_big = L.adell_conv3d_bwd_weight_f16x3_workspace(ctypes.byref(d_big))
    L.adell_set_tuning(b"wgrad_nozring", 1)
    try:
        ws_big_generic = L.adell_conv3d_bwd_weight_f16x3_workspace(ctypes.byref(d_big))
    finally:
        L.adell_set_tuning(b"wgrad_nozring", 0)
    assert ws_big > ws_big_generic          # one slab per block of the sub-lattice kernel
    assert L.adell_conv3d_bwd_weight_f16x3_workspace(ctypes.byref(d_small)) > 0
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 32, 64, 128, 128, generator=g)
    dy = torch.randn(2, 32, 32, 64, 64, generator=g) * 1e-2
    dw, db = ops.conv3d_bwd_weight(ops.ndhwc(x.to(cuda)), ops.ndhwc(dy.to(cuda)), 3, 2, 1,
                                   want_db=True, f16x3=True)
    wr = torch.zeros(32, 32, 3, 3, 3, dtype=torch.float64, requires_grad=True)
    torch.nn.functional.conv3d(x.double(), wr, None, stride=2, padding=1).backward(dy.double())
    assert _rel(dw.view(32, 32, 3, 3, 3).cpu().double(), wr.grad) < 5e-6
    assert _rel(db.cpu().double(), dy.double().sum((0, 2, 3, 4))) < 5e-6
