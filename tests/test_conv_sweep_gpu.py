"""Seeded random sweep of the convolution layer (forward, dX for both concat sources, dW, db)
against stock torch fp32 on the CPU: kernel sizes 1 / 3 / 5, strides 1 / 2 (per axis), paddings,
ragged volumes, virtual concat, residual input, channel counts that hit every launch plan (small
Cin, multiples of 16 and 32, ragged tiles, the two-wave strided instance, the z-ring weight
gradient with 16-channel pieces, split-K levels). Tolerance: 2e-5 of the output scale (f16x3 carries
22 bits per product; torch's own CPU conv is the fp32 reference of the same op)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from adell_mri_amd import functional as HF
from adell_mri_amd import ops

pytestmark = pytest.mark.gpu


def _cases(n, seed):
    rng = np.random.default_rng(seed)
    chans = [1, 2, 3, 4, 8, 16, 24, 32, 48, 64, 80]
    out = []
    while len(out) < n:
        k = int(rng.choice([1, 3, 3, 3, 5]))
        stride = tuple(int(v) for v in rng.choice([1, 1, 2], size=3))
        pad = tuple(int(rng.integers(0, k // 2 + 1)) for _ in range(3))
        size = tuple(int(rng.integers(max(k, 4), 21)) for _ in range(3))
        if rng.random() < 0.25:
            size = tuple(int(v) for v in rng.choice([8, 16, 24, 32], size=3))
        c0 = int(rng.choice(chans))
        c1 = int(rng.choice([0, 0, 0, 16, 32, c0]))
        if c1 and (c0 % 4 or c1 % 4):
            c1 = 0
        cout = int(rng.choice(chans))
        n_items = int(rng.choice([1, 1, 2, 3]))
        osz = tuple((s + 2 * p - k) // st + 1 for s, p, st in zip(size, pad, stride))
        if min(osz) < 1:
            continue
        residual = bool(rng.random() < 0.2 and stride == (1, 1, 1) and osz == size)
        out.append((n_items, c0, c1, cout, size, k, stride, pad, residual))
    return out


def _rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-20))


@pytest.mark.parametrize("case", _cases(48, 20260202), ids=lambda c: "n{}_c{}+{}_o{}_{}x{}x{}_k{}_s{}{}{}_p{}{}{}_r{}".format(
    c[0], c[1], c[2], c[3], *c[4], c[5], *c[6], *c[7], int(c[8])))
def test_conv_layer_matches_torch_cpu(cuda, case):
    n_items, c0, c1, cout, size, k, stride, pad, residual = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x0 = torch.randn(n_items, c0, *size, generator=g)
    x1 = torch.randn(n_items, c1, *size, generator=g) * 0.5 if c1 else None
    w = torch.randn(cout, c0 + c1, k, k, k, generator=g) / np.sqrt((c0 + c1) * k ** 3)
    b = torch.randn(cout, generator=g)
    xin = x0 if x1 is None else torch.cat([x0, x1], 1)
    ref_in = [t.clone().requires_grad_(True) for t in (xin, w, b)]
    y_ref = F.conv3d(ref_in[0], ref_in[1], ref_in[2], stride=stride, padding=pad)
    res = torch.randn(y_ref.shape, generator=g) if residual else None
    if residual:
        y_ref = y_ref + res
    r = torch.randn(y_ref.shape, generator=g)
    (y_ref * r).sum().backward()

    hx0 = ops.ndhwc(x0.to(cuda)).requires_grad_(True)
    hx1 = ops.ndhwc(x1.to(cuda)).requires_grad_(True) if c1 else None
    hw, hb = w.to(cuda).requires_grad_(True), b.to(cuda).requires_grad_(True)
    y = HF.conv3d(hx0, hw, hb, stride, pad, x1=hx1,
                  residual=None if res is None else ops.ndhwc(res.to(cuda)))
    assert tuple(y.shape) == tuple(y_ref.shape)
    (y * ops.ndhwc(r.to(cuda))).sum().backward()
    tol = 2e-5
    assert _rel(y.detach().cpu(), y_ref.detach()) < tol
    dx_ref = ref_in[0].grad
    assert _rel(hx0.grad.cpu(), dx_ref[:, :c0]) < tol
    if c1:
        assert _rel(hx1.grad.cpu(), dx_ref[:, c0:]) < tol
    assert _rel(hw.grad.cpu(), ref_in[1].grad) < tol
    assert _rel(hb.grad.cpu(), ref_in[2].grad) < tol
    # statistics by-product of the forward epilogue (the fused norm reads them)
    part = getattr(y, "_adell_partials", None)
    if part is not None and part.numel() > 0:
        V = int(np.prod(y.shape[2:]))
        mean, _ = ops.stats_finalize(part, V, 1e-5)
        want = y.detach().flatten(2).mean(-1)
        assert float((mean - want).abs().max()) < 1e-4 * float(want.abs().max() + 1.0)
