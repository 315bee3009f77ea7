"""The specialised (3^3, stride 1, 8x8x4 brick) instance of the f16x3 implicit-GEMM kernel against
the generic instance and the fp32-MFMA kernel, at sizes that have interior and ragged bricks."""
import os

import pytest
import torch

from adell_mri_amd import _lib, ops

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


@pytest.mark.parametrize("cin,c1,cout,size,res", [(32, 0, 32, (40, 44, 52), False),
                                                    (64, 0, 64, (48, 40, 36), True),
                                                    (32, 16, 32, (36, 40, 48), True),
                                                    (64, 0, 32, (64, 64, 32), False),
                                                    (32, 0, 96, (40, 40, 40), False)])
def test_spec_instance_matches_generic_and_fp32(cuda, cin, c1, cout, size, res):
    g = torch.Generator().manual_seed(cin + cout)
    D, H, W = size
    c0 = cin - c1
    x0 = ops.ndhwc((torch.randn(2, c0, D, H, W, generator=g) * 3).to(cuda))
    x1 = ops.ndhwc(torch.randn(2, c1, D, H, W, generator=g).to(cuda)) if c1 else None
    w = (torch.randn(cout, cin, 3, 3, 3, generator=g) * 0.05).to(cuda)
    b = torch.randn(cout, generator=g).to(cuda)
    r = ops.ndhwc(torch.randn(2, cout, D, H, W, generator=g).to(cuda)) if res else None
    wp = ops.pack_weight_f16x3(w, 0)

    def run():
        return ops.conv3d_fwd(x0, wp, b, cout, 3, 1, 1, x1=x1, residual=r, want_stats=True)

    y_spec, st_spec = run()
    with _lib.tuning(igemm_nospec=1):
        y_gen, st_gen = run()
    y32, _ = ops.conv3d_fwd(x0, ops.pack_weight(w, 0), b, cout, 3, 1, 1, x1=x1, residual=r,
                            want_stats=True)
    assert _rel(y_spec, y_gen) < 2e-6
    assert _rel(y_spec, y32) < 5e-6
    # per-channel (sum, sum of squares) partials [N, tiles, Cout, 2]: same totals whichever
    # brick-to-block mapping wrote them, and equal to the sums of the output itself
    t_spec, t_gen = st_spec.double().sum(1), st_gen.double().sum(1)
    assert _rel(t_spec, t_gen) < 1e-6
    yd = y_spec.double()
    want = torch.stack([yd.sum((2, 3, 4)), (yd * yd).sum((2, 3, 4))], -1)
    assert _rel(t_spec, want) < 1e-5


@pytest.mark.parametrize("cin,cout,size,split", [(32, 32, (40, 44, 52), 0), (64, 32, (40, 40, 48), 0),
                                                  (64, 32, (40, 40, 40), 32)])
def test_spec_backward_data_matches_generic(cuda, cin, cout, size, split):
    """dX of a 3^3 stride-1 conv runs the same instance on dY with flipped taps (and a split
    store into the two sources of a virtual concat)."""
    g = torch.Generator().manual_seed(7)
    D, H, W = size
    dy = ops.ndhwc((torch.randn(1, cout, D, H, W, generator=g) * 1e-3).to(cuda))
    w = (torch.randn(cout, cin, 3, 3, 3, generator=g) * 0.05).to(cuda)
    wpb = ops.pack_weight_f16x3(w, 1)

    def run():
        return ops.conv3d_bwd_data(dy, wpb, (D, H, W), cin - split, split, 3, 1, 1)

    a = run()
    with _lib.tuning(igemm_nospec=1):
        b = run()
    a = a if isinstance(a, (tuple, list)) else (a,)
    b = b if isinstance(b, (tuple, list)) else (b,)
    for u, v in zip(a, b):
        if u is not None:
            assert _rel(u, v) < 2e-6


@pytest.mark.parametrize("n,c0,c1,cout,size,res", [(1, 128, 0, 128, (8, 8, 8), True),
                                                   (2, 64, 64, 64, (16, 16, 16), False),
                                                   (1, 256, 0, 128, (16, 16, 16), False),
                                                   (1, 128, 0, 256, (6, 10, 12), True)])
def test_split_k_matches_single_pass(cuda, monkeypatch, n, c0, c1, cout, size, res):
    """Low-resolution layers share the channel chunks of a brick out over several blocks and fold
    the slabs (csrc/conv3d.hip): same y, same statistics partials, same split dX as one pass."""
    g = torch.Generator().manual_seed(c0 + cout)
    D, H, W = size
    x0 = ops.ndhwc(torch.randn(n, c0, D, H, W, generator=g).to(cuda))
    x1 = ops.ndhwc(torch.randn(n, c1, D, H, W, generator=g).to(cuda)) if c1 else None
    w = (torch.randn(cout, c0 + c1, 3, 3, 3, generator=g) * 0.05).to(cuda)
    b = torch.randn(cout, generator=g).to(cuda)
    r = ops.ndhwc(torch.randn(n, cout, D, H, W, generator=g).to(cuda)) if res else None
    dy = ops.ndhwc((torch.randn(n, cout, D, H, W, generator=g) * 1e-3).to(cuda))
    wp, wpb = ops.pack_weight_f16x3(w, 0), ops.pack_weight_f16x3(w, 1)
    d = ops.make_conv_desc(n, size, c0, c1, cout, 3, 1, 1)
    import ctypes
    assert _lib.lib().adell_conv3d_splitk_workspace(ctypes.byref(d), 0) > 0   # the case does split

    def run():
        y, st = ops.conv3d_fwd(x0, wp, b, cout, 3, 1, 1, x1=x1, residual=r, want_stats=True)
        dx = ops.conv3d_bwd_data(dy, wpb, size, c0, c1, 3, 1, 1)
        return y, st, dx

    y_s, st_s, dx_s = run()
    with _lib.tuning(no_splitk=1):
        y_1, st_1, dx_1 = run()
    assert _rel(y_s, y_1) < 2e-6
    assert _rel(st_s.double().sum(1), st_1.double().sum(1)) < 1e-5
    for u, v in zip(dx_s, dx_1):
        if u is not None:
            assert _rel(u, v) < 2e-6
    y_s2, _, _ = run()
    assert torch.equal(y_s, y_s2)   # fixed fold order


@pytest.mark.parametrize("n,cin,cout,size,pad", [(1, 32, 32, (16, 24, 32), 1), (2, 16, 48, (12, 8, 20), 1),
                                                 (1, 64, 64, (8, 8, 8), 0), (1, 32, 32, (64, 64, 64), 1)])
def test_stride2_backward_data_by_parity_classes(cuda, monkeypatch, n, cin, cout, size, pad):
    """dX of a stride-2 k = 3 conv computed as 8 stride-1 sub-kernel convs on the stride-2 lattice
    (adell_conv3d_bwd_data_s2_f16x3) == the zero-insertion formulation == torch."""
    from adell_mri_amd import functional as HF
    monkeypatch.setitem(HF.FLAGS, "s2class_always", True)   # also below the size where it pays
    g = torch.Generator().manual_seed(cin + size[0])
    x = torch.randn(n, cin, *size, generator=g)
    w = torch.randn(cout, cin, 3, 3, 3, generator=g) * 0.05

    def run():
        xd = ops.ndhwc(x.to(cuda)).requires_grad_(True)
        wd = w.to(cuda).requires_grad_(True)
        y = HF.conv3d(xd, wd, None, stride=2, padding=pad)
        gy = torch.Generator().manual_seed(1)
        dy = torch.randn(y.shape, generator=gy).to(cuda)
        y.backward(ops.ndhwc(dy))
        return y.detach(), dy, xd.grad.detach()

    y, dy, dx_c = run()
    monkeypatch.setitem(HF.FLAGS, "no_s2class", True)
    _, _, dx_z = run()
    xr = x.double().requires_grad_(True)
    yr = torch.nn.functional.conv3d(xr, w.double(), None, stride=2, padding=pad)
    yr.backward(dy.cpu().double())
    assert _rel(dx_c.cpu().double(), xr.grad) < 5e-6
    assert _rel(dx_c, dx_z) < 5e-6
