"""The streaming factor-2 transposed-conv kernels (csrc/convt_k2.hip: forward, dX, dW on the fp32
MFMA) against stock torch fp32 on the CPU: 32 / 64 channels on either side, ragged last tile of 32
voxels, batch > 1; and the layer dispatch (large volumes take this path, small ones and other
channel counts the implicit-GEMM kernels)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from adell_mri_amd import functional as HF
from adell_mri_amd import ops

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-20))


@pytest.mark.parametrize("n,cin,cout,size", [(1, 32, 32, (32, 32, 32)), (2, 64, 32, (20, 33, 25)),
                                              (1, 32, 64, (17, 40, 49)), (1, 64, 64, (32, 32, 33)),
                                              # 16 output channels: the (fx, co) column forms
                                              (2, 32, 16, (20, 33, 25)), (1, 64, 16, (32, 32, 33))])
@pytest.mark.parametrize("factors", [(2, 2, 2), (2, 2, 1)])
def test_convt_k2_matches_torch_cpu(cuda, n, cin, cout, size, factors):
    if factors == (2, 2, 1) and cout == 16:
        pytest.skip("16 output channels: factors 2 x 2 x 2 only")
    g = torch.Generator().manual_seed(cin + cout + size[0])
    x = torch.randn(n, cin, *size, generator=g).requires_grad_(True)
    w = (torch.randn(cin, cout, *factors, generator=g) / np.sqrt(cin)).requires_grad_(True)
    b = torch.randn(cout, generator=g).requires_grad_(True)
    y_ref = F.conv_transpose3d(x, w, b, stride=factors)
    r = torch.randn(y_ref.shape, generator=g)
    (y_ref * r).sum().backward()
    hx = ops.ndhwc(x.detach().to(cuda)).requires_grad_(True)
    hw, hb = w.detach().to(cuda).requires_grad_(True), b.detach().to(cuda).requires_grad_(True)
    assert ops.convt_k2_ok(hx.shape, hw)
    ops.KERNEL_TIMER = ops.KernelTimer()
    try:
        y = HF.conv_transpose3d(hx, hw, hb)
        (y * ops.ndhwc(r.to(cuda))).sum().backward()
        names = ops.KERNEL_TIMER.summary()
    finally:
        ops.KERNEL_TIMER = None
    assert names["adell_convt_k2_kernel"]["launches"] == 3
    assert _rel(y.detach().cpu(), y_ref.detach()) < 2e-6
    assert _rel(hx.grad.cpu(), x.grad) < 2e-6
    assert _rel(hw.grad.cpu(), w.grad) < 5e-6
    assert _rel(hb.grad.cpu(), b.grad) < 5e-5     # by-product of the dW kernel (fp32 sums of ~10^5 terms)
    # deterministic weight gradient (fixed fold order of the block partials)
    dw2 = ops.convt_k2_bwd_weight(hx.detach(), ops.ndhwc(r.to(cuda)), factors=factors)
    assert torch.equal(dw2, hw.grad)


def test_convt_k2_dispatch_limits(cuda):
    w = torch.zeros(32, 32, 2, 2, 2, device=cuda)
    assert ops.convt_k2_ok((2, 32, 32, 32, 16), w)            # 32 768 voxels
    assert not ops.convt_k2_ok((1, 32, 16, 16, 16), w)        # too small: implicit GEMM
    assert not ops.convt_k2_ok((1, 48, 64, 64, 64), torch.zeros(48, 32, 2, 2, 2, device=cuda))
    assert ops.convt_k2_ok((1, 32, 64, 64, 64), torch.zeros(32, 32, 2, 2, 1, device=cuda))   # width kept
    assert not ops.convt_k2_ok((1, 32, 64, 64, 64), torch.zeros(32, 32, 1, 2, 2, device=cuda))
    assert not ops.convt_k2_ok((1, 32, 64, 64, 64), torch.zeros(32, 16, 2, 2, 1, device=cuda))
    ops.FLAGS["no_convt_k2"] = True
    try:
        assert not ops.convt_k2_ok((2, 32, 32, 32, 16), w)
    finally:
        ops.FLAGS["no_convt_k2"] = False
