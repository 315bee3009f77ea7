"""The persistent wave-specialised f16x3 instance (csrc/conv_igemm_ws.h) against the
one-brick-per-block instances and the fp64 C oracle: forward with virtual concat / bias / residual /
statistics partials, backward-data with the split store, ragged and interior bricks, more and
fewer work items than blocks."""
import numpy as np
import pytest
import torch

from adell_mri_amd import _lib, ops
from oracle import cops

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


@pytest.mark.parametrize("n,cin,c1,cout,size,res", [
    (2, 64, 0, 64, (40, 44, 52), True),      # 64-channel tile, 8x8x4 bricks, ragged everywhere
    (1, 32, 0, 32, (40, 40, 48), False),     # 32-channel tile, 8x8x8 bricks
    (2, 32, 16, 32, (24, 28, 36), True),     # virtual concat of two sources
    (1, 64, 0, 32, (64, 64, 32), False),
    (1, 32, 0, 64, (16, 24, 16), True),      # fewer bricks than blocks
    (1, 128, 64, 64, (32, 32, 32), False),   # 8 chunks, two sources
    (3, 16, 0, 32, (16, 16, 24), True),      # one chunk per brick
    (1, 32, 0, 96, (24, 24, 24), False),     # two column tiles of the 64-channel instance
])
def test_ws_forward_matches_block_instances(cuda, n, cin, c1, cout, size, res):
    g = torch.Generator().manual_seed(cin + cout + size[0])
    D, H, W = size
    c0 = cin - c1
    x0 = ops.ndhwc((torch.randn(n, c0, D, H, W, generator=g) * 3).to(cuda))
    x1 = ops.ndhwc(torch.randn(n, c1, D, H, W, generator=g).to(cuda)) if c1 else None
    w = (torch.randn(cout, cin, 3, 3, 3, generator=g) * 0.05).to(cuda)
    b = torch.randn(cout, generator=g).to(cuda)
    r = ops.ndhwc(torch.randn(n, cout, D, H, W, generator=g).to(cuda)) if res else None
    wp = ops.pack_weight_f16x3(w, 0)

    def run():
        amax = torch.zeros(1, device=cuda, dtype=torch.int32)
        y, st = ops.conv3d_fwd(x0, wp, b, cout, 3, 1, 1, x1=x1, residual=r, want_stats=True,
                               amax=amax)
        return y, st, amax

    with _lib.tuning(ws_min_items=1, igemm_ws=1):
        y_ws, st_ws, am_ws = run()
        y_ws2, _, _ = run()
    with _lib.tuning(igemm_ws=0):
        y_bk, st_bk, am_bk = run()
    assert torch.equal(y_ws, y_ws2)                       # deterministic
    assert _rel(y_ws, y_bk) < 2e-6
    assert int(am_ws) == int(am_bk)                       # absmax by-product of the input
    t_ws, t_bk = st_ws.double().sum(1), st_bk.double().sum(1)
    assert st_ws.shape == st_bk.shape and _rel(t_ws, t_bk) < 1e-6
    yd = y_ws.double()
    want = torch.stack([yd.sum((2, 3, 4)), (yd * yd).sum((2, 3, 4))], -1)
    assert _rel(t_ws, want) < 1e-5


@pytest.mark.parametrize("cin,cout,size,split", [(32, 32, (40, 44, 52), 0), (64, 32, (40, 40, 48), 0),
                                                  (64, 32, (40, 40, 40), 32), (64, 64, (24, 20, 28), 0)])
def test_ws_backward_data_matches_block_instances(cuda, cin, cout, size, split):
    g = torch.Generator().manual_seed(7)
    D, H, W = size
    dy = ops.ndhwc((torch.randn(2, cout, D, H, W, generator=g) * 1e-3).to(cuda))
    w = (torch.randn(cout, cin, 3, 3, 3, generator=g) * 0.05).to(cuda)
    wpb = ops.pack_weight_f16x3(w, 1)

    def run():
        return ops.conv3d_bwd_data(dy, wpb, (D, H, W), cin - split, split, 3, 1, 1)

    with _lib.tuning(ws_min_items=1, igemm_ws=1):
        a = run()
    with _lib.tuning(igemm_ws=0):
        b = run()
    for u, v in zip(a, b):
        if u is not None:
            assert _rel(u, v) < 2e-6


@pytest.mark.parametrize("cin,cout", [(32, 32), (64, 64)])
def test_ws_against_c_oracle(cuda, cin, cout):
    shape = (24, 20, 28)
    rng = np.random.default_rng(cin)
    x = rng.standard_normal((2, cin, *shape), dtype=np.float32)
    w = (rng.standard_normal((cout, cin, 3, 3, 3), dtype=np.float32) * 0.05).astype(np.float32)
    b = rng.standard_normal((cout,), dtype=np.float32)
    y_ref = cops.conv3d(x, w, b, 1, 1)
    xt, wt, bt = ops.ndhwc(torch.from_numpy(x).to(cuda)), torch.from_numpy(w).to(cuda), torch.from_numpy(b).to(cuda)
    with _lib.tuning(ws_min_items=1, igemm_ws=1):
        y, _ = ops.conv3d_fwd(xt, ops.pack_weight_f16x3(wt, 0), bt, cout, 3, 1, 1)
    assert np.abs(y.cpu().numpy() - y_ref).max() / np.abs(y_ref).max() < 2e-6
