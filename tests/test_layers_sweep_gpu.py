"""Seeded random sweeps against stock torch fp32 on the CPU of (a) the fused norm -> activation
pass (instance / batch / no normalisation, affine or not, every activation of the factory) with its
backward, and (b) the transposed convolution (kernel == stride, factors 1 / 2 per axis) with dX, dW
and db: ragged volumes, channel counts on both sides of every vector-width / tile boundary."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from adell_mri_amd import functional as HF
from adell_mri_amd import ops

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-20))


ACTS = {"identity": lambda t, w: t, "swish": lambda t, w: F.silu(t), "relu": lambda t, w: F.relu(t),
        "leaky_relu": lambda t, w: F.leaky_relu(t, 0.01), "prelu": lambda t, w: F.prelu(t, w),
        "gelu": lambda t, w: F.gelu(t), "sigmoid": lambda t, w: torch.sigmoid(t),
        "tanh": lambda t, w: torch.tanh(t), "elu": lambda t, w: F.elu(t)}


def _adn_cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        out.append((int(rng.choice([1, 2, 3])), int(rng.choice([1, 2, 3, 4, 6, 8, 16, 24, 32, 64, 128])),
                    tuple(int(rng.integers(2, 14)) for _ in range(3)),
                    str(rng.choice(["instance", "batch", "none"])), bool(rng.random() < 0.6),
                    str(rng.choice(list(ACTS)))))
    return out


@pytest.mark.parametrize("case", _adn_cases(36, 7), ids=lambda c: "n{}_c{}_{}x{}x{}_{}_aff{}_{}".format(
    c[0], c[1], *c[2], c[3], int(c[4]), c[5]))
def test_norm_act_matches_torch_cpu(cuda, case):
    n, c, size, norm, affine, act = case
    if norm == "instance" and int(np.prod(size)) < 2:
        pytest.skip("instance norm of one voxel")
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = (torch.randn(n, c, *size, generator=g) * 1.7 + 0.3).requires_grad_(True)
    gamma = (torch.rand(c, generator=g) + 0.5).requires_grad_(True) if affine else None
    beta = (torch.randn(c, generator=g) * 0.2).requires_grad_(True) if affine else None
    aw = (torch.rand(c if c % 2 else 1, generator=g) * 0.3).requires_grad_(True) if act == "prelu" else None
    if norm == "instance":
        h = F.instance_norm(x, weight=gamma, bias=beta, eps=1e-5)
    elif norm == "batch":
        h = F.batch_norm(x, None, None, weight=gamma, bias=beta, training=True, eps=1e-5)
    else:
        h = x if gamma is None else x * gamma.view(1, c, 1, 1, 1) + beta.view(1, c, 1, 1, 1)
    y_ref = ACTS[act](h, aw)
    r = torch.randn(y_ref.shape, generator=g)
    (y_ref * r).sum().backward()

    hx = ops.ndhwc(x.detach().to(cuda)).requires_grad_(True)
    hg = gamma.detach().to(cuda).requires_grad_(True) if affine else None
    hb = beta.detach().to(cuda).requires_grad_(True) if affine else None
    haw = aw.detach().to(cuda).requires_grad_(True) if aw is not None else None
    y = HF.norm_drop_act(hx, norm=norm, eps=1e-5, gamma=hg, beta=hb, act=act,
                         act_p={"leaky_relu": 0.01, "elu": 1.0}.get(act, 0.0),
                         act_w=haw, training=True)
    (y * ops.ndhwc(r.to(cuda))).sum().backward()
    assert _rel(y.detach().cpu(), y_ref.detach()) < 2e-5
    assert _rel(hx.grad.cpu(), x.grad) < 2e-4
    if affine:
        assert _rel(hg.grad.cpu(), gamma.grad) < 2e-4 and _rel(hb.grad.cpu(), beta.grad) < 2e-4
    if aw is not None:
        assert _rel(haw.grad.cpu(), aw.grad) < 2e-4


def _convt_cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        f = tuple(int(v) for v in rng.choice([1, 2, 2], size=3))
        if f == (1, 1, 1):
            f = (2, 2, 1)
        out.append((int(rng.choice([1, 2])), int(rng.choice([4, 8, 16, 24, 32, 64])),
                    int(rng.choice([4, 8, 16, 32, 48, 64])),
                    tuple(int(rng.integers(2, 11)) for _ in range(3)), f))
    return out


@pytest.mark.parametrize("case", _convt_cases(16, 11), ids=lambda c: "n{}_c{}_o{}_{}x{}x{}_f{}{}{}".format(
    c[0], c[1], c[2], *c[3], *c[4]))
def test_transposed_conv_matches_torch_cpu(cuda, case):
    n, cin, cout, size, f = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = torch.randn(n, cin, *size, generator=g).requires_grad_(True)
    w = (torch.randn(cin, cout, *f, generator=g) / np.sqrt(cin)).requires_grad_(True)
    b = torch.randn(cout, generator=g).requires_grad_(True)
    y_ref = F.conv_transpose3d(x, w, b, stride=f)
    r = torch.randn(y_ref.shape, generator=g)
    (y_ref * r).sum().backward()
    hx = ops.ndhwc(x.detach().to(cuda)).requires_grad_(True)
    hw, hb = w.detach().to(cuda).requires_grad_(True), b.detach().to(cuda).requires_grad_(True)
    y = HF.conv_transpose3d(hx, hw, hb)
    assert tuple(y.shape) == tuple(y_ref.shape)
    (y * ops.ndhwc(r.to(cuda))).sum().backward()
    assert _rel(y.detach().cpu(), y_ref.detach()) < 2e-5
    assert _rel(hx.grad.cpu(), x.grad) < 2e-5
    assert _rel(hw.grad.cpu(), w.grad) < 2e-5
    assert _rel(hb.grad.cpu(), b.grad) < 2e-5
