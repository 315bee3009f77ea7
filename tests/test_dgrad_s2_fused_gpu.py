"""One-launch backward-data of the U-Net downsampling layer (32 -> 32 channels, k = 3, stride 2,
padding 1; unet.py:571-579): csrc/conv_dgrad_s2.hip against torch's fp64 conv backward and against
the eight-launch parity-class path, on whole and ragged bricks, with and without the parked fork
gradient, plus the absmax by-product."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)


@pytest.mark.parametrize("n,size", [(2, (16, 16, 16)), (1, (12, 20, 10)), (1, (8, 8, 72)),
                                    (3, (2, 2, 2)), (1, (64, 64, 64))])
@pytest.mark.parametrize("with_add", [False, True])
def test_fused_backward_data_matches_fp64_and_the_class_path(cuda, n, size, with_add):
    from adell_mri_amd import functional as HF
    from adell_mri_amd import ops

    g = torch.Generator().manual_seed(size[0] + n)
    w = torch.randn(32, 32, 3, 3, 3, generator=g) * 0.05
    osz = ops.conv_out_size(size, (3,) * 3, (2,) * 3, (1,) * 3)
    dy = torch.randn(n, 32, *osz, generator=g) * torch.rand(n, 1, *osz, generator=g) * 3.0
    add0 = torch.randn(n, 32, *size, generator=g)
    assert ops.conv3d_bwd_data_s2_fused_ok(size, 32, 0, 32, (3,) * 3, (2,) * 3, (1,) * 3)
    wd = w.to(cuda)
    amax = torch.zeros(1, dtype=torch.int32, device=cuda)
    got = ops.conv3d_bwd_data_s2_fused(ops.ndhwc(dy.to(cuda)), HF._packed(wd, 1), size, amax=amax,
                                       add0=ops.ndhwc(add0.to(cuda)) if with_add else None)
    xr = torch.zeros(n, 32, *size, dtype=torch.float64, requires_grad=True)
    torch.nn.functional.conv3d(xr, w.double(), None, stride=2, padding=1).backward(dy.double())
    want = xr.grad + (add0.double() if with_add else 0.0)
    assert _rel(got.cpu().double(), want) < 5e-6
    classes = ops.conv3d_bwd_data_s2(ops.ndhwc(dy.to(cuda)), HF._packed_s2_classes(wd, (1, 1, 1)),
                                     size, 32, (1, 1, 1))
    if with_add:
        classes = classes + add0.to(cuda)
    assert _rel(got, classes) < 5e-6
    assert amax.view(torch.float32).item() == float(dy.abs().max())


def test_not_applicable_outside_its_layer(cuda):
    from adell_mri_amd import ops

    ok = ops.conv3d_bwd_data_s2_fused_ok
    assert not ok((16, 16, 16), 64, 0, 32, (3,) * 3, (2,) * 3, (1,) * 3)
    assert not ok((16, 16, 16), 32, 0, 64, (3,) * 3, (2,) * 3, (1,) * 3)
    assert not ok((16, 16, 15), 32, 0, 32, (3,) * 3, (2,) * 3, (1,) * 3)
    assert not ok((16, 16, 16), 32, 0, 32, (3,) * 3, (2,) * 3, (0,) * 3)
    assert not ok((16, 16, 16), 32, 0, 32, (3,) * 3, (1,) * 3, (1,) * 3)
    assert not ok((16, 16, 16), 16, 16, 32, (3,) * 3, (2,) * 3, (1,) * 3)


def test_autograd_takes_the_fused_kernel(cuda, monkeypatch):
    from adell_mri_amd import functional as HF
    from adell_mri_amd import ops

    calls = []
    real = ops.conv3d_bwd_data_s2_fused
    monkeypatch.setattr(ops, "conv3d_bwd_data_s2_fused",
                        lambda *a, **k: calls.append(1) or real(*a, **k))
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 32, 16, 16, 16, generator=g)
    w = torch.randn(32, 32, 3, 3, 3, generator=g) * 0.05
    dy = torch.randn(1, 32, 8, 8, 8, generator=g)
    xd = ops.ndhwc(x.to(cuda)).requires_grad_(True)
    wd = w.to(cuda).requires_grad_(True)
    HF.conv3d(xd, wd, None, stride=2, padding=1).backward(ops.ndhwc(dy.to(cuda)))
    assert calls == [1]
    xr = x.double().requires_grad_(True)
    wr = w.double().requires_grad_(True)
    torch.nn.functional.conv3d(xr, wr, None, stride=2, padding=1).backward(dy.double())
    assert _rel(xd.grad.cpu().double(), xr.grad) < 5e-6
    assert _rel(wd.grad.cpu().double(), wr.grad) < 5e-6


def test_fused_pair_against_the_c_oracle(cuda):
    """Forward and backward-data of the layer against the fp64-accumulated C oracle (oracle/c), on
    a shape with whole and ragged bricks."""
    import numpy as np

    from adell_mri_amd import functional as HF
    from adell_mri_amd import ops
    from oracle import cops

    rng = np.random.default_rng(5)
    size = (12, 20, 18)
    x = rng.standard_normal((1, 32, *size)).astype(np.float32)
    x[:, ::3] *= 1e-3
    w = (rng.standard_normal((32, 32, 3, 3, 3)) / np.sqrt(32 * 27)).astype(np.float32)
    b = rng.standard_normal(32).astype(np.float32)
    dy = (rng.standard_normal((1, 32, 6, 10, 9)) * 1e-4).astype(np.float32)
    y_ref = cops.conv3d(x, w, b, 2, 1)
    dx_ref, _, _ = cops.conv3d_bwd(x, w, dy, 2, 1)
    wd = torch.from_numpy(w).to(cuda)
    y, _ = ops.conv3d_fwd(ops.ndhwc(torch.from_numpy(x).to(cuda)), HF._packed(wd, 0),
                          torch.from_numpy(b).to(cuda), 32, 3, 2, 1)
    dx = ops.conv3d_bwd_data_s2_fused(ops.ndhwc(torch.from_numpy(dy).to(cuda)), HF._packed(wd, 1), size)
    assert float(np.abs(y.cpu().numpy() - y_ref).max() / np.abs(y_ref).max()) < 5e-6
    assert float(np.abs(dx.cpu().numpy() - dx_ref).max() / np.abs(dx_ref).max()) < 5e-6
