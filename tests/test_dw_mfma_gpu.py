"""Depthwise 7^3 convolution as Toeplitz products on the f16x3 MFMA (csrc/dw_mfma.hip): forward and
backward-data against torch fp64 on the CPU and against the vector-ALU kernels it replaces, with
ragged planes, both column forms (register column for D <= 16, streamed otherwise), badly scaled
operands (the per-(item, channel) power-of-two scales have to carry them) and the dispatch limits."""
import pytest
import torch
import torch.nn.functional as F

from adell_mri_amd import _lib, functional as HF, ops

pytestmark = pytest.mark.gpu

CASES = [  # N, C, (D, H, W)
    (2, 8, (16, 16, 16)),
    (3, 12, (10, 11, 13)),      # ragged rows and columns, D not a multiple of the group of 4
    (1, 4, (21, 9, 16)),        # streamed column (D > 16), last group ragged
    (1, 20, (3, 16, 9)),        # fewer planes than taps
    (2, 4, (40, 12, 12)),
]


def _rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.mark.parametrize("N,C,size", CASES)
@pytest.mark.parametrize("scale", [1.0, 2e4, 3e-5])
def test_forward_and_backward_data_against_fp64(cuda, N, C, size, scale):
    g = torch.Generator().manual_seed(C + size[0])
    x = torch.randn(N, C, *size, generator=g, dtype=torch.float64) * scale
    x[:, 1] *= 300.0                       # channels of one block far apart: per-channel scales
    x[0, :, size[0] // 2] *= 50.0          # one plane far above the others
    w = torch.randn(C, 1, 7, 7, 7, generator=g, dtype=torch.float64) * 0.05
    w[2] *= 1e-3
    b = torch.randn(C, generator=g, dtype=torch.float64) * scale
    xr = x.clone().requires_grad_(True)
    y = F.conv3d(xr, w, b, padding=3, groups=C)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(dy)
    xd, wd, bd = ops.ndhwc(x.float().to(cuda)), w.float().to(cuda), b.float().to(cuda)
    dyd = ops.ndhwc(dy.float().to(cuda))
    assert _lib.lib().adell_dw_mfma_ok(N, C, *size, 7, 7, 7, xd.data_ptr(), xd.data_ptr())
    got_y = ops.dwconv3d_fwd(xd, wd, bd)
    got_dx = ops.dwconv3d_bwd_data(dyd, wd)
    # per channel: the split carries 22 bits relative to the channel's own largest value
    for c in range(C):
        assert _rel(got_y[:, c].cpu().double(), y[:, c].detach()) < 2e-6, c
        assert _rel(got_dx[:, c].cpu().double(), xr.grad[:, c]) < 2e-6, c
    with _lib.tuning(dw_nomfma=1):
        old_y = ops.dwconv3d_fwd(xd, wd, bd)
        old_dx = ops.dwconv3d_bwd_data(dyd, wd)
    assert _rel(got_y, old_y) < 2e-6 and _rel(got_dx, old_dx) < 2e-6
    # deterministic
    assert torch.equal(got_y, ops.dwconv3d_fwd(xd, wd, bd))


def test_tiny_operands_give_tiny_results_not_garbage(cuda):
    """Operands of 1e-30: the power-of-two scales of the two operands are undone one after the
    other (their sum leaves the exponent range of a float: 2^-(kx + kw) built as ONE bit pattern
    wrapped into garbage / NaN). Forward, backward-data, weight gradient and the dense small-volume
    form: finite, and equal to fp64 within the split's 22 bits wherever fp32 can hold the value."""
    g = torch.Generator().manual_seed(5)
    for size, N in (((16, 16, 16), 2), ((4, 4, 4), 8)):
        x = torch.randn(N, 8, *size, generator=g, dtype=torch.float64) * 1e-30
        w = torch.randn(8, 1, 7, 7, 7, generator=g, dtype=torch.float64) * 1e-6
        y = F.conv3d(x, w, None, padding=3, groups=8)           # ~1e-35: a normal fp32 number
        xd, wd = ops.ndhwc(x.float().to(cuda)), w.float().to(cuda)
        got = ops.dwconv3d_fwd(xd, wd, None)
        assert torch.isfinite(got).all()
        assert _rel(got.cpu().double(), y) < 1e-4
        got_dx = ops.dwconv3d_bwd_data(xd, wd)                    # the same product, flipped taps
        assert torch.isfinite(got_dx).all() and float(got_dx.abs().max()) < 1e-30
    x = torch.randn(2, 8, 16, 16, 16, generator=g) * 1e-30
    dw, db = ops.dwconv3d_bwd_weight(ops.ndhwc(x.to(cuda)), ops.ndhwc((x * 1e-5).to(cuda)), (7, 7, 7), True)
    assert torch.isfinite(dw).all() and torch.isfinite(db).all() and float(dw.abs().max()) < 1e-30


def test_dispatch_limits_and_the_fp32_mode(cuda):
    ok = _lib.lib().adell_dw_mfma_ok
    p = torch.zeros(64, device=cuda).data_ptr()
    assert ok(2, 96, 16, 16, 16, 7, 7, 7, p, p)
    assert not ok(2, 96, 16, 16, 16, 5, 5, 5, p, p)       # other stencils: vector-ALU kernels
    assert not ok(2, 96, 16, 8, 8, 7, 7, 7, p, p)         # rows of <= 8 voxels: a quarter of the tile
    assert not ok(2, 96, 16, 16, 17, 7, 7, 7, p, p)       # a row must fit the 16 columns
    assert not ok(2, 98, 16, 16, 16, 7, 7, 7, p, p)       # channels in fours
    assert not ok(2, 96, 16, 16, 16, 7, 7, 7, p + 4, p)   # 16-byte aligned tensors
    # "fp32" precision keeps the depthwise convs on exact fp32 FMAs too
    old = HF.CONV_PRECISION
    try:
        HF.set_conv_precision("fp32")
        assert not ok(2, 96, 16, 16, 16, 7, 7, 7, p, p)
        HF.set_conv_precision("f16x3")
        assert ok(2, 96, 16, 16, 16, 7, 7, 7, p, p)
    finally:
        HF.set_conv_precision(old)


@pytest.mark.parametrize("N,C,size", CASES + [(9, 8, (6, 16, 16))])   # several items per chunk, D < 7
@pytest.mark.parametrize("scale", [1.0, 2e4, 3e-5])
def test_weight_and_bias_gradient_against_fp64(cuda, N, C, size, scale):
    """csrc/dw_wgrad_mfma.hip: rows as the reduction dimension of f16x3 MFMA products, one power-of-two
    scale per tensor -- against torch fp64 and the vector-ALU tile kernel."""
    g = torch.Generator().manual_seed(C + size[1])
    x = torch.randn(N, C, *size, generator=g, dtype=torch.float64) * scale
    x[0, :, size[0] // 2] *= 30.0
    dy = torch.randn(N, C, *size, generator=g, dtype=torch.float64) / scale
    dy[:, 1] *= 20.0
    w = torch.zeros(C, 1, 7, 7, 7, dtype=torch.float64, requires_grad=True)
    b = torch.zeros(C, dtype=torch.float64, requires_grad=True)
    F.conv3d(x, w, b, padding=3, groups=C).backward(dy)
    xd, dyd = ops.ndhwc(x.float().to(cuda)), ops.ndhwc(dy.float().to(cuda))
    assert _lib.lib().adell_dw_wgrad_mfma_ok(N, C, *size, 7, 7, 7, xd.data_ptr(), dyd.data_ptr())
    dw, db = ops.dwconv3d_bwd_weight(xd, dyd, (7, 7, 7), True)
    assert _rel(dw.cpu().double(), w.grad) < 3e-6
    assert _rel(db.cpu().double(), b.grad) < 3e-6
    with _lib.tuning(dw_wgrad_nomfma=1):
        dw_old, db_old = ops.dwconv3d_bwd_weight(xd, dyd, (7, 7, 7), True)
    assert _rel(dw, dw_old) < 3e-6 and _rel(db, db_old) < 3e-6
    dw2, db2 = ops.dwconv3d_bwd_weight(xd, dyd, (7, 7, 7), True)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)      # deterministic: chunk order fold
    dw3, _ = ops.dwconv3d_bwd_weight(xd, dyd, (7, 7, 7), False)
    assert torch.equal(dw, dw3)


@pytest.mark.parametrize("N,C,size", [(64, 8, (4, 4, 4)), (5, 12, (4, 4, 4)), (17, 4, (3, 4, 2)), (33, 8, (4, 2, 4)),
                                      (40, 32, (4, 4, 4)), (3, 16, (3, 3, 4))])
@pytest.mark.parametrize("scale", [1.0, 2e4, 3e-5])
def test_small_volumes_as_a_dense_matrix(cuda, N, C, size, scale):
    """csrc/dw_dense.hip: volumes of at most 4^3 voxels -- every output sees every input -- as a
    [items] x [64] x [64] product per channel; forward and backward-data against fp64 and the
    vector-ALU kernels, ragged item tiles and ragged volumes."""
    g = torch.Generator().manual_seed(N + C)
    x = torch.randn(N, C, *size, generator=g, dtype=torch.float64) * scale
    x[:, 1] *= 300.0
    w = torch.randn(C, 1, 7, 7, 7, generator=g, dtype=torch.float64) * 0.05
    b = torch.randn(C, generator=g, dtype=torch.float64) * scale
    xr = x.clone().requires_grad_(True)
    y = F.conv3d(xr, w, b, padding=3, groups=C)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(dy)
    xd, wd, bd = ops.ndhwc(x.float().to(cuda)), w.float().to(cuda), b.float().to(cuda)
    dyd = ops.ndhwc(dy.float().to(cuda))
    assert _lib.lib().adell_dw_dense_ok(N, C, *size, 7, 7, 7, xd.data_ptr(), xd.data_ptr())
    got_y, got_dx = ops.dwconv3d_fwd(xd, wd, bd), ops.dwconv3d_bwd_data(dyd, wd)
    for c in range(C):
        assert _rel(got_y[:, c].cpu().double(), y[:, c].detach()) < 2e-6, c
        assert _rel(got_dx[:, c].cpu().double(), xr.grad[:, c]) < 2e-6, c
    with _lib.tuning(dw_nomfma=1):
        old_y, old_dx = ops.dwconv3d_fwd(xd, wd, bd), ops.dwconv3d_bwd_data(dyd, wd)
    assert _rel(got_y, old_y) < 2e-6 and _rel(got_dx, old_dx) < 2e-6
    assert torch.equal(got_y, ops.dwconv3d_fwd(xd, wd, bd))
    # 2^3 and 5 x 4 x 4 volumes are not this kernel's
    p = xd.data_ptr()
    assert not _lib.lib().adell_dw_dense_ok(N, C, 2, 2, 2, 7, 7, 7, p, p)
    assert not _lib.lib().adell_dw_dense_ok(N, C, 5, 4, 4, 7, 7, 7, p, p)
