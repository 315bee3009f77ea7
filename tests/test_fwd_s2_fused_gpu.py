"""One-launch forward of the U-Net downsampling layer (32 -> 32 channels, k = 3, stride 2, padding 1;
unet.py:571-579): csrc/conv_fwd_s2.hip against torch's fp64 conv and the implicit-GEMM path, on
whole and ragged bricks, with bias, statistics partials and the absmax by-product; autograd through
HF.conv3d takes it."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)


@pytest.mark.parametrize("n,size", [(2, (16, 16, 16)), (1, (12, 20, 10)), (1, (8, 8, 72)),
                                    (3, (2, 2, 2)), (1, (64, 64, 64))])
@pytest.mark.parametrize("bias", [False, True])
def test_fused_forward_matches_fp64_and_the_igemm_path(cuda, monkeypatch, n, size, bias):
    from adell_mri_amd import functional as HF
    from adell_mri_amd import ops

    g = torch.Generator().manual_seed(size[1] + n)
    x = torch.randn(n, 32, *size, generator=g) * (torch.rand(n, 1, *size, generator=g) * 4.0 + 0.01)
    w = torch.randn(32, 32, 3, 3, 3, generator=g) * 0.05
    b = torch.randn(32, generator=g) if bias else None
    want = torch.nn.functional.conv3d(x.double(), w.double(), None if b is None else b.double(),
                                      stride=2, padding=1)
    xd, wd = ops.ndhwc(x.to(cuda)), w.to(cuda)
    bd = None if b is None else b.to(cuda)
    pack = HF._packed(wd, 0)
    amax = torch.zeros(1, dtype=torch.int32, device=cuda)
    y, part = ops.conv3d_fwd(xd, pack, bd, 32, 3, 2, 1, want_stats=True, amax=amax)
    assert _rel(y.cpu().double(), want) < 5e-6
    stats = part.double().sum(1).cpu()
    ref = torch.stack([want.sum((2, 3, 4)), (want ** 2).sum((2, 3, 4))], -1)
    assert _rel(stats, ref) < 1e-5
    assert amax.view(torch.float32).item() == float(x.abs().max())
    monkeypatch.setitem(ops.FLAGS, "no_s2fused", True)
    y2, _ = ops.conv3d_fwd(xd, pack, bd, 32, 3, 2, 1, want_stats=False)
    assert _rel(y, y2) < 5e-6


def test_chunks_of_very_different_magnitude(cuda):
    """The eight sub-lattices of a brick get their own operand scale: make them differ by many
    orders of magnitude so that the accumulators are rescaled between them."""
    from adell_mri_amd import functional as HF
    from adell_mri_amd import ops

    g = torch.Generator().manual_seed(9)
    x = torch.randn(1, 32, 16, 16, 16, generator=g)
    mag = torch.tensor([1e-5, 3e3])
    x = x * mag[torch.arange(16) % 2].view(1, 1, 16, 1, 1) * mag[torch.arange(16) % 2].view(1, 1, 1, 1, 16)
    w = torch.randn(32, 32, 3, 3, 3, generator=g) * 0.05
    want = torch.nn.functional.conv3d(x.double(), w.double(), None, stride=2, padding=1)
    y, _ = ops.conv3d_fwd(ops.ndhwc(x.to(cuda)), HF._packed(w.to(cuda), 0), None, 32, 3, 2, 1)
    scale = torch.nn.functional.conv3d(x.double().abs(), w.double().abs(), None, stride=2, padding=1).max()
    assert float((y.cpu().double() - want).abs().max() / scale) < 2e-6


def test_autograd_step_through_the_fused_pair(cuda, monkeypatch):
    from adell_mri_amd import _lib
    from adell_mri_amd import functional as HF
    from adell_mri_amd import ops

    seen = []
    real = _lib.lib().adell_conv3d_fwd_s2_fused

    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 32, 16, 16, 16, generator=g)
    w = torch.randn(32, 32, 3, 3, 3, generator=g) * 0.05
    b = torch.randn(32, generator=g)
    dy = torch.randn(2, 32, 8, 8, 8, generator=g)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    torch.nn.functional.conv3d(xr, wr, br, stride=2, padding=1).backward(dy.double())
    xd = ops.ndhwc(x.to(cuda)).requires_grad_(True)
    wd, bd = w.to(cuda).requires_grad_(True), b.to(cuda).requires_grad_(True)
    timer = ops.KernelTimer()
    ops.KERNEL_TIMER = timer
    try:
        y = HF.conv3d(xd, wd, bd, stride=2, padding=1)
        y.backward(ops.ndhwc(dy.to(cuda)))
    finally:
        ops.KERNEL_TIMER = None
    names = {name for (name, _tag) in timer.by_tag()}
    assert "adell_fwd_s2_fused_kernel" in names and "adell_dgrad_s2_fused_kernel" in names
    assert _rel(xd.grad.cpu().double(), xr.grad) < 5e-6
    assert _rel(wd.grad.cpu().double(), wr.grad) < 5e-6
    assert _rel(bd.grad.cpu().double(), br.grad) < 5e-6
    del seen, real
