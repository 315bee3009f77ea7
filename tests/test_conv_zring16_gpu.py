"""The z-marching 16 -> 16 channel convolution (csrc/conv_zring16.hip, v_mfma_f32_16x16x32_f16):
forward and backward-data against torch fp64 on the CPU and against the implicit-GEMM instances it
replaces (``igemm_no16``), with bias, residual, statistics partials, the absmax by-product, ragged
planes, padding 0 / 1, badly scaled operands and split-row sources."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

from adell_mri_amd import _lib, ops

pytestmark = pytest.mark.gpu

CASES = [  # N, size, padding
    (2, (16, 24, 40), 1),
    (1, (9, 17, 33), 1),       # ragged planes, odd depth
    (1, (14, 18, 22), 0),
    (3, (8, 8, 8), 1),         # one column per item
    (1, (48, 40, 24), 1),      # several z segments
]


def _rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.mark.parametrize("N,size,pad", CASES)
@pytest.mark.parametrize("scale", [1.0, 3e4, 2e-5])
def test_forward_against_fp64_and_the_replaced_instances(cuda, N, size, pad, scale):
    g = torch.Generator().manual_seed(size[0] + 7 * pad)
    x = torch.randn(N, 16, *size, generator=g) * scale
    x[:, :, size[0] // 2] *= 40.0              # one plane far above the others: per-plane scales
    w = torch.randn(16, 16, 3, 3, 3, generator=g) * 0.1
    b = torch.randn(16, generator=g)
    ref = F.conv3d(x.double(), w.double(), b.double(), padding=pad)
    res = torch.randn(ref.shape, generator=g) * scale
    ref = ref + res.double()
    hx, hres = ops.ndhwc(x.to(cuda)), ops.ndhwc(res.to(cuda))
    wp = ops.pack_weight_f16x3(w.to(cuda), 0)
    amax = torch.zeros(1, device=cuda, dtype=torch.int32)
    y, part = ops.conv3d_fwd(hx, wp, b.to(cuda), 16, 3, 1, pad, residual=hres, want_stats=True, amax=amax)
    assert _rel(y.cpu().double(), ref) < 2e-6
    # statistics partials: per-channel sum and sum of squares of what was stored
    s = part.sum(dim=1).cpu().double()
    yd = y.cpu().double()
    assert torch.allclose(s[..., 0], yd.sum(dim=(2, 3, 4)), rtol=1e-4, atol=1e-4 * float(yd.abs().max()))
    assert torch.allclose(s[..., 1], (yd * yd).sum(dim=(2, 3, 4)), rtol=1e-4)
    # the by-product the weight gradient reads: absmax of the input tensor
    assert float(amax.view(torch.float32)) == float(x.abs().max())
    with _lib.tuning(igemm_no16=1):
        y_old, part_old = ops.conv3d_fwd(hx, wp, b.to(cuda), 16, 3, 1, pad, residual=hres, want_stats=True)
    assert _rel(y, y_old) < 2e-6
    # deterministic
    y2, part2 = ops.conv3d_fwd(hx, wp, b.to(cuda), 16, 3, 1, pad, residual=hres, want_stats=True)
    assert torch.equal(y, y2) and torch.equal(part, part2)


def test_the_plan_is_the_new_kernels(cuda):
    """4 x 96^3 (UNETR's full-resolution level): columns x segments rows per item, not bricks."""
    d = ops.make_conv_desc(4, (96, 96, 96), 16, 0, 16, 3, 1, 1)
    rows = _lib.lib().adell_conv3d_fwd_ntiles_f16x3(ctypes.byref(d))
    with _lib.tuning(igemm_no16=1):
        bricks = _lib.lib().adell_conv3d_fwd_ntiles_f16x3(ctypes.byref(d))
    assert bricks == 12 * 12 * 12 and rows % 144 == 0 and rows != bricks
    # other channel counts keep their plans
    d2 = ops.make_conv_desc(4, (96, 96, 96), 32, 0, 16, 3, 1, 1)
    a = _lib.lib().adell_conv3d_fwd_ntiles_f16x3(ctypes.byref(d2))
    with _lib.tuning(igemm_no16=1):
        assert a == _lib.lib().adell_conv3d_fwd_ntiles_f16x3(ctypes.byref(d2))


@pytest.mark.parametrize("N,size,pad", CASES[:3])
def test_backward_data_against_fp64(cuda, N, size, pad):
    g = torch.Generator().manual_seed(size[1] + pad)
    x = torch.randn(N, 16, *size, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(16, 16, 3, 3, 3, generator=g, dtype=torch.float64) * 0.1
    y = F.conv3d(x, w, padding=pad)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64) * 1e-3
    y.backward(dy)
    wp = ops.pack_weight_f16x3(w.float().to(cuda), 1)
    hdy = ops.ndhwc(dy.float().to(cuda))
    dx = ops.conv3d_bwd_data(hdy, wp, size, 16, 0, 3, 1, pad)[0]
    assert _rel(dx.cpu().double(), x.grad) < 2e-6
    add = ops.ndhwc(torch.randn(N, 16, *size, generator=g).to(cuda))
    dx2 = ops.conv3d_bwd_data(hdy, wp, size, 16, 0, 3, 1, pad, add0=add)[0]
    assert _rel(dx2.cpu().double(), x.grad + add.cpu().double()) < 2e-6


@pytest.mark.parametrize("N,size,pad", CASES[:2])
def test_split_row_source(cuda, N, size, pad):
    g = torch.Generator().manual_seed(3)
    x = ops.ndhwc((torch.randn(N, 16, *size, generator=g) * 1.5).to(cuda))
    w = (torch.randn(16, 16, 3, 3, 3, generator=g) * 0.05).to(cuda)
    b = torch.randn(16, generator=g).to(cuda)
    assert ops.conv3d_rows_ok(N, size, 16, 0, 16, 3, 1, pad)
    wp = ops.pack_weight_f16x3(w, 0)
    r0, s0 = ops.rows_from_f32(x, 9)
    v0 = ops.rows_to_f32(r0, s0)                 # the values the rows hold
    y_ref, p_ref = ops.conv3d_fwd(v0, wp, b, 16, 3, 1, pad, want_stats=True)
    before = ops.ROWS_FALLBACKS[0]
    y, p = ops.conv3d_fwd(r0, wp, b, 16, 3, 1, pad, want_stats=True, rows0=s0)
    assert ops.ROWS_FALLBACKS[0] == before
    assert _rel(y, y_ref) < 2e-6
    assert torch.allclose(p.sum(1), p_ref.sum(1), rtol=1e-4, atol=1e-3)
