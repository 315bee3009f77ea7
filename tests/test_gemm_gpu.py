"""fp32-MFMA row-major GEMM (csrc/gemm.hip) behind torch.nn.Linear: every operand layout,
ragged sizes, split-k and the fused bias / residual epilogue against float64 numpy."""
import numpy as np
import pytest
import torch

from adell_mri_amd import functional as HF
from adell_mri_amd import ops


def rel(a, r):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else a
    return float(np.abs(a - r).max() / (np.abs(r).max() + 1e-30))


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (7, 5, 3), (16, 1024, 768), (33, 130, 50),
                                   (128, 768, 3072), (300, 96, 384), (1024, 384, 1536),
                                   (4099, 97, 96), (2, 2, 70000), (65, 257, 129),
                                   # ViT of UNETR (64 x 64 tiles, no split) / 128 x 128 tiles
                                   (864, 512, 512), (864, 1536, 512), (4100, 520, 72),
                                   # a handful of features over many rows: the streaming kernels
                                   (70000, 8, 2), (70001, 2, 8), (66000, 32, 8), (65540, 12, 5),
                                   (2, 8, 70000), (8, 32, 20000), (32, 32, 16400), (5, 3, 33000),
                                   # round 5: four outputs per thread / N K up to 512 (rows kernel);
                                   # the LDS-staged weight-gradient form (dense power-of-two M, N)
                                   (70000, 8, 32), (66000, 64, 8), (65540, 8, 64), (70000, 16, 32),
                                   (8, 2, 70000), (64, 8, 16388), (8, 64, 65600), (32, 8, 100000),
                                   (16, 16, 16384), (2, 4, 16388)])
@pytest.mark.parametrize("layout", ["nt", "nn", "tn"])
def test_gemm_layouts_match_float64(cuda, M, N, K, layout):
    rng = np.random.default_rng(M * 131 + N * 17 + K)
    a = rng.standard_normal((M, K))
    b = rng.standard_normal((K, N))
    ref = a @ b
    A = torch.from_numpy(a.astype(np.float32)).to(cuda)
    B = torch.from_numpy(b.astype(np.float32)).to(cuda)
    if layout == "nt":    # forward: X [M, K], W [N, K]
        out = ops.gemm(M, N, K, A.contiguous(), K, True, B.t().contiguous(), K, True)
    elif layout == "nn":  # backward-data: dY [M, K], W [K, N]
        out = ops.gemm(M, N, K, A.contiguous(), K, True, B.contiguous(), N, False)
    else:                 # backward-weight: dY^T stored [K, M], X [K, N]
        out = ops.gemm(M, N, K, A.t().contiguous(), M, False, B.contiguous(), N, False)
    refs = a.astype(np.float32).astype(np.float64) @ b.astype(np.float32).astype(np.float64)
    tol = 2e-6 * max(1.0, np.sqrt(K) / 8)
    assert rel(out, refs) < tol, (rel(out, refs), rel(out, ref))


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(5, 7, 9), (432, 768, 768), (16, 2048, 1024), (2048, 96, 384)])
def test_linear_bias_residual_and_grads(cuda, M, N, K):
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn((M, K), generator=g, dtype=torch.float64).requires_grad_(True)
    w = (torch.randn((N, K), generator=g, dtype=torch.float64) / np.sqrt(K)).requires_grad_(True)
    b = torch.randn((N,), generator=g, dtype=torch.float64).requires_grad_(True)
    r = torch.randn((M, N), generator=g, dtype=torch.float64).requires_grad_(True)
    y = torch.nn.functional.linear(x, w, b) + r
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(dy)
    xd, wd, bd, rd = [t.detach().float().to(cuda).requires_grad_(True) for t in (x, w, b, r)]
    yd = HF.linear(xd, wd, bd, residual=rd)
    yd.backward(dy.float().to(cuda))
    assert rel(yd, y.detach().numpy()) < 5e-6
    assert rel(xd.grad, x.grad.numpy()) < 5e-6
    assert rel(wd.grad, w.grad.numpy()) < 5e-6
    assert rel(bd.grad, b.grad.numpy()) < 5e-6
    assert rel(rd.grad, r.grad.numpy()) < 1e-7


@pytest.mark.gpu
def test_linear_on_token_tensor_keeps_leading_shape(cuda):
    x = torch.randn((2, 24, 32), device=cuda, requires_grad=True)
    w = torch.randn((48, 32), device=cuda, requires_grad=True)
    y = HF.linear(x, w)
    assert y.shape == (2, 24, 48)
    ref = x.detach().double().cpu() @ w.detach().double().cpu().t()
    assert rel(y, ref.numpy()) < 5e-6
    y.sum().backward()
    assert x.grad.shape == x.shape and w.grad.shape == w.shape
