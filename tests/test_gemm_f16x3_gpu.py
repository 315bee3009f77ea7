"""f16x3 GEMM (csrc/gemm_f16x3.hip): the three operand layouts of a Linear layer (forward X W^T,
backward-data dY W, backward-weight dY^T X), ragged tiles, split-K, bias / residual epilogue and
extreme operand scales, against fp64; HF.linear takes it when ops.FLAGS["gemm_f16x3"] is set (the
default, for outputs of at least 64 x 64: ops.FLAGS["gemm_f16x3_min_mn"]) and falls back to the
fp32-MFMA GEMM when the operands do not qualify."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _opt_in(monkeypatch):
    from adell_mri_amd import ops

    monkeypatch.setitem(ops.FLAGS, "gemm_f16x3", True)
    monkeypatch.setitem(ops.FLAGS, "gemm_f16x3_min_mn", 0)     # every size: the kernel is under test
    monkeypatch.setitem(ops.FLAGS, "gemm_f16x3_min_k", 0)


def _rel(a, b):
    return float((a.detach().cpu().double() - b).abs().max() / b.abs().max())


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (256, 384, 96), (100, 36, 64), (4, 8, 4),
                                   (1000, 132, 260), (64, 3072, 768), (4096, 96, 384)])
@pytest.mark.parametrize("a_kc,b_kc", [(True, True), (True, False), (False, False), (False, True)])
@pytest.mark.parametrize("words", [False, True])
def test_layouts_against_fp64(cuda, M, N, K, a_kc, b_kc, words):
    from adell_mri_amd import ops

    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g)
    B = torch.randn(K, N, generator=g)
    bias = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g)
    want = A.double() @ B.double() + bias.double() + res.double()
    Ad = (A if a_kc else A.t()).contiguous().to(cuda)
    Bd = (B.t() if b_kc else B).contiguous().to(cuda)
    lda, ldb = (K if a_kc else M), (K if b_kc else N)
    if not ops.gemm_f16x3_ok(M, N, K, Ad, lda, a_kc, Bd, ldb, b_kc):
        assert (not a_kc and M % 4) or (not b_kc and N % 4) or K % 4
        pytest.skip("operands do not qualify (checked: the reason is an extent that is not a multiple of 4)")
    # operand scales: chosen per block and stage inside the kernel, or per tensor from absmax words
    wa, wb = (ops.absmax_word(Ad), ops.absmax_word(Bd)) if words else (None, None)
    got = ops.gemm_f16x3(M, N, K, Ad, lda, a_kc, Bd, ldb, b_kc, wa, wb, bias=bias.to(cuda),
                         residual=res.to(cuda))
    scale = (A.double().abs() @ B.double().abs()).max()
    assert float((got.cpu().double() - want).abs().max() / scale) < 2e-6


def test_split_k_is_deterministic_and_exact_enough(cuda):
    from adell_mri_amd import _lib, ops

    M, N, K = 96, 160, 65536       # dW of a point-wise layer: K = rows
    assert _lib.lib().adell_gemm_f16x3_workspace_floats(M, N, K) > 0
    g = torch.Generator().manual_seed(1)
    A = torch.randn(K, M, generator=g).to(cuda)     # outer-contiguous operands
    B = torch.randn(K, N, generator=g).to(cuda)
    c1 = ops.gemm_f16x3(M, N, K, A, M, False, B, N, False)
    c2 = ops.gemm_f16x3(M, N, K, A, M, False, B, N, False)
    assert torch.equal(c1, c2)
    want = A.cpu().double().t() @ B.cpu().double()
    assert _rel(c1, want) < 3e-6


@pytest.mark.parametrize("sa,sb", [(1e-7, 1.0), (3e6, 1e-5), (1.0, 1e-12)])
def test_operand_scales(cuda, sa, sb):
    from adell_mri_amd import ops

    g = torch.Generator().manual_seed(3)
    A = (torch.randn(200, 128, generator=g) * sa).to(cuda)
    B = (torch.randn(64, 128, generator=g) * sb).to(cuda)
    want = A.cpu().double() @ B.cpu().double().t()
    got = ops.gemm_f16x3(200, 64, 128, A, 128, True, B, 128, True, ops.absmax_word(A), ops.absmax_word(B))
    assert _rel(got, want) < 3e-6
    assert _rel(ops.gemm_f16x3(200, 64, 128, A, 128, True, B, 128, True), want) < 3e-6


def test_scale_changes_between_stages_rescale_the_accumulators(cuda):
    """K blocks of very different magnitude: the per-stage exponents change along K and the
    accumulators follow by exact powers of two."""
    from adell_mri_amd import ops

    g = torch.Generator().manual_seed(11)
    M, N, K = 130, 140, 512
    A = torch.randn(M, K, generator=g)
    B = torch.randn(N, K, generator=g)
    mag = torch.tensor([1e-6, 1.0, 3e4, 1e-3, 1.0, 1e5, 1e-8, 1.0]).repeat_interleave(64)
    A, B = A * mag, B * mag.flip(0)
    want = A.double() @ B.double().t()
    got = ops.gemm_f16x3(M, N, K, A.to(cuda), K, True, B.to(cuda), K, True)
    scale = (A.double().abs() @ B.double().abs().t()).max()
    assert float((got.cpu().double() - want).abs().max() / scale) < 2e-6


def test_linear_takes_the_f16x3_gemm_and_matches_torch(cuda, monkeypatch):
    from adell_mri_amd import functional as HF
    from adell_mri_amd import ops

    calls = []
    real = ops.gemm_f16x3
    monkeypatch.setattr(ops, "gemm_f16x3", lambda *a, **k: calls.append(a[:3]) or real(*a, **k))
    g = torch.Generator().manual_seed(5)
    x = torch.randn(3, 40, 96, generator=g)
    w = torch.randn(384, 96, generator=g) / 10
    b = torch.randn(384, generator=g)
    r = torch.randn(3, 40, 384, generator=g)
    dy = torch.randn(3, 40, 384, generator=g)
    xr, wr, br, rr = (t.double().requires_grad_(True) for t in (x, w, b, r))
    (torch.nn.functional.linear(xr, wr, br) + rr).backward(dy.double())
    xd, wd, bd, rd = (t.to(cuda).requires_grad_(True) for t in (x, w, b, r))
    y = HF.linear(xd, wd, bd, residual=rd)
    y.backward(dy.to(cuda))
    assert calls == [(120, 384, 96), (120, 96, 384), (384, 96, 120)]
    assert _rel(y, (torch.nn.functional.linear(xr, wr, br) + rr).detach()) < 2e-6
    assert _rel(xd.grad, xr.grad) < 2e-6 and _rel(wd.grad, wr.grad) < 2e-6
    assert _rel(bd.grad, br.grad) < 2e-6 and torch.equal(rd.grad.cpu(), dy)
    # 5 input features: no 16-byte rows -> the fp32-MFMA GEMM
    calls.clear()
    x5 = torch.randn(7, 5, generator=g).to(cuda).requires_grad_(True)
    w5 = torch.randn(12, 5, generator=g).to(cuda).requires_grad_(True)
    HF.linear(x5, w5).sum().backward()
    assert calls == [] and x5.grad is not None and w5.grad is not None
    # the default size rule: narrow outputs (SWIN's 24 ... 48-wide projections) stay on the fp32 GEMM
    monkeypatch.setitem(ops.FLAGS, "gemm_f16x3_min_mn", 64)
    xn = torch.randn(3, 40, 96, generator=g).to(cuda).requires_grad_(True)
    wn = (torch.randn(48, 96, generator=g) / 10).to(cuda).requires_grad_(True)
    HF.linear(xn, wn).sum().backward()
    assert calls == []          # (the forward's shape decides for the layer's three GEMMs)
    # and not at all when switched off
    monkeypatch.setitem(ops.FLAGS, "gemm_f16x3", False)
    HF.linear(xd, wd, bd).sum().backward()
    assert calls == []


@pytest.mark.parametrize("rows,k,hid,n,act", [(4096, 96, 384, 96, "gelu"), (300, 192, 768, 192, "gelu"),
                                               (128, 2048, 128, 64, "swish"),      # split K: the fold applies it
                                               (1000, 64, 256, 128, "leaky_relu")])
def test_mlp_with_the_activation_in_the_gemm_epilogues(cuda, monkeypatch, rows, k, hid, n, act):
    """functional.mlp (Linear -> act -> Linear + residual, activation and its backward inside GEMM
    epilogues) against the layer-by-layer form and torch fp64 on the CPU."""
    import torch.nn.functional as F
    from adell_mri_amd import functional as HF
    g = torch.Generator().manual_seed(rows + k)
    x = torch.randn(rows, k, generator=g).to(cuda).requires_grad_(True)
    w1 = (torch.randn(hid, k, generator=g) * k ** -0.5).to(cuda).requires_grad_(True)
    b1 = (torch.randn(hid, generator=g) * 0.3).to(cuda).requires_grad_(True)
    w2 = (torch.randn(n, hid, generator=g) * hid ** -0.5).to(cuda).requires_grad_(True)
    b2 = torch.randn(n, generator=g).to(cuda).requires_grad_(True)
    res = torch.randn(rows, n, generator=g).to(cuda).requires_grad_(True)
    dy = torch.randn(rows, n, generator=g).to(cuda)
    leaves = [x, w1, b1, w2, b2, res]
    slope = 0.02 if act == "leaky_relu" else 0.0
    assert not HF.mlp_ok(x, w1, w2)                       # opt-in (measured slower per step)
    monkeypatch.setitem(HF.FLAGS, "no_mlp_fuse", False)
    assert HF.mlp_ok(x, w1, w2) == (rows * hid >= HF.FLAGS["mlp_min_elems"])   # the size rule
    monkeypatch.setitem(HF.FLAGS, "mlp_min_elems", 0)
    assert HF.mlp_ok(x, w1, w2)

    def run(fused):
        for t in leaves:
            t.grad = None
        if fused:
            y = HF.mlp(x, w1, b1, w2, b2, act=act, act_p=slope, residual=res)
        else:
            hdn = HF.elementwise(HF.linear(x, w1, b1), act=act, act_p=slope)
            y = HF.linear(hdn, w2, b2, residual=res)
        y.backward(dy)
        torch.cuda.synchronize()
        return [y.detach().clone()] + [t.grad.clone() for t in leaves]

    new, old = run(True), run(False)
    fn = {"gelu": F.gelu, "swish": F.silu, "leaky_relu": lambda t: F.leaky_relu(t, slope)}[act]
    ld = [t.detach().cpu().double().requires_grad_(True) for t in leaves]
    yd = fn(ld[0] @ ld[1].T + ld[2]) @ ld[3].T + ld[4] + ld[5]
    yd.backward(dy.cpu().double())
    ref = [yd.detach()] + [t.grad for t in ld]
    for name, u, v, r in zip(["y", "dx", "dw1", "db1", "dw2", "db2", "dres"], new, old, ref):
        scale = float(r.abs().max())
        assert float((u.cpu().double() - r).abs().max()) < 2e-5 * scale, name
        assert float((v.cpu().double() - r).abs().max()) < 2e-5 * scale, name
    # deterministic
    again = run(True)
    assert all(torch.equal(a, b) for a, b in zip(new, again))


def test_mlp_module_takes_the_fused_path_only_without_norm_or_dropout(cuda, monkeypatch):
    """linear_blocks.MLP: Linear -> ADN -> Linear runs on functional.mlp when the ADN is a bare
    activation; a LayerNorm or an active dropout inside it keeps the layer-by-layer form."""
    from adell_mri_amd import functional as HF
    from adell_mri_amd.modules.layers.adn_fn import get_adn_fn
    from adell_mri_amd.modules.layers.linear_blocks import MLP
    monkeypatch.setitem(HF.FLAGS, "mlp_min_elems", 0)
    monkeypatch.setitem(HF.FLAGS, "no_mlp_fuse", False)
    calls = []
    orig = HF.mlp
    monkeypatch.setattr(HF, "mlp", lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
    torch.manual_seed(3)
    x = torch.randn(2, 200, 64, device=cuda, requires_grad=True)
    gy = torch.randn(2, 200, 64, device=cuda)
    for norm, drop, fused in (("identity", 0.0, True), ("layer", 0.0, False), ("identity", 0.3, False)):
        m = MLP(64, 64, [256], adn_fn=get_adn_fn(1, norm, "gelu", drop)).to(cuda).train()
        calls.clear()
        y = m(x, residual=x)
        assert bool(calls) == fused, (norm, drop)
        if not fused or drop:
            continue
        y.backward(gy)
        got = [y.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in m.parameters()]
        x.grad = None
        for p in m.parameters():
            p.grad = None
        monkeypatch.setitem(HF.FLAGS, "no_mlp_fuse", True)
        y2 = m(x, residual=x)
        y2.backward(gy)
        monkeypatch.setitem(HF.FLAGS, "no_mlp_fuse", False)
        want = [y2.detach(), x.grad] + [p.grad for p in m.parameters()]
        for u, v in zip(got, want):
            assert float((u - v).abs().max()) < 2e-5 * float(v.abs().max())
        x.grad = None
    # in eval mode the dropout is inactive: bare activation again
    m = MLP(64, 64, [256], adn_fn=get_adn_fn(1, "identity", "gelu", 0.3)).to(cuda).eval()
    calls.clear()
    with torch.no_grad():
        m(x)
    assert calls


# ---- the streaming kernel for many-row Linear layers (csrc/gemm_rows.hip) --------------------------
ROWS_CASES = [  # M, N, K: persistent blocks over (128-row tile, 128-column slice) pairs
    (65536, 384, 96),       # ConvNeXt pwconv1 (a quarter of config 4's rows): 3 slices x 3 stages
    (65536, 96, 384),       # pwconv2: one slice of 3 column tiles, 12 stages
    (66000, 200, 64),       # ragged last row tile, last slice of 72 columns (a partial column tile)
    (70001, 36, 32),        # N < one slice, odd M, a single stage
    (40000, 1536, 384),     # 12 slices
]


@pytest.mark.parametrize("M,N,K", ROWS_CASES)
@pytest.mark.parametrize("b_kc", [True, False])
def test_rows_kernel_against_fp64_and_the_tile_kernel(cuda, M, N, K, b_kc):
    from adell_mri_amd import _lib, ops

    g = torch.Generator().manual_seed(M % 1000 + N + K)
    A = torch.randn(M, K, generator=g)
    A[:, : K // 2] *= 40.0                      # block exponents differ between k stages
    A[M // 3] *= 3000.0                         # and between row blocks
    B = torch.randn(K, N, generator=g) * torch.logspace(-3, 2, N)[None, :]   # per-column scales matter
    bias = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g)
    Ad, Bd = A.to(cuda), (B.t() if b_kc else B).contiguous().to(cuda)
    ldb = K if b_kc else N
    assert _lib.lib().adell_gemm_f16x3_workspace_floats(M, N, K) >= 128 * ((N + 127) // 128) * (K + 1)
    got = ops.gemm_f16x3(M, N, K, Ad, K, True, Bd, ldb, b_kc, bias=bias.to(cuda), residual=res.to(cuda))
    with _lib.tuning(gemm_norows=1):
        old = ops.gemm_f16x3(M, N, K, Ad, K, True, Bd, ldb, b_kc, bias=bias.to(cuda), residual=res.to(cuda))
    # row by row against fp64 in chunks (the whole fp64 product of the largest case is 0.5 GB)
    worst, worst_old = 0.0, 0.0
    Bd64, absB = B.double(), B.double().abs()
    for lo in range(0, M, 16384):
        a64 = A[lo:lo + 16384].double()
        want = a64 @ Bd64 + bias.double() + res[lo:lo + 16384].double()
        scale = (a64.abs() @ absB).clamp_min(1e-30)             # per element: sum |a||b|
        worst = max(worst, float(((got[lo:lo + 16384].cpu().double() - want).abs() / scale).max()))
        worst_old = max(worst_old, float(((old[lo:lo + 16384].cpu().double() - want).abs() / scale).max()))
    assert worst < 2e-6, (worst, worst_old)
    assert torch.equal(got, ops.gemm_f16x3(M, N, K, Ad, K, True, Bd, ldb, b_kc, bias=bias.to(cuda),
                                           residual=res.to(cuda)))            # deterministic


def test_rows_kernel_activation_pair(cuda):
    """Linear -> GELU with both tensors from one launch, and the backward of the pair (C * act'(saved))
    on the streaming kernel, against torch fp64 and the tile kernel's epilogue."""
    from adell_mri_amd import _lib, ops

    M, K, N = 50000, 96, 384
    g = torch.Generator().manual_seed(11)
    A = torch.randn(M, K, generator=g).to(cuda)
    W = (torch.randn(N, K, generator=g) * 0.1).to(cuda)
    b = torch.randn(N, generator=g).to(cuda)
    pre, post = ops.gemm_f16x3_act(M, N, K, A, K, True, W, K, True, "gelu", bias=b, want_act=True)
    want = A.cpu().double() @ W.cpu().double().t() + b.cpu().double()
    assert _rel(pre, want) < 2e-6
    assert _rel(post, torch.nn.functional.gelu(want)) < 2e-6
    # backward of the pair: dX = (dY W2) * gelu'(pre), A = dY [M, K2], W2 [K2, N] outer-contiguous
    K2 = 96
    dY = torch.randn(M, K2, generator=g).to(cuda)
    W2 = (torch.randn(K2, N, generator=g) * 0.1).to(cuda)
    dh, _ = ops.gemm_f16x3_act(M, N, K2, dY, K2, True, W2, N, False, "gelu", dact_in=pre)
    x = want.clone().requires_grad_(True)
    torch.nn.functional.gelu(x).backward(dY.cpu().double() @ W2.cpu().double())
    assert _rel(dh, x.grad) < 3e-6
    with _lib.tuning(gemm_norows=1):
        pre_o, post_o = ops.gemm_f16x3_act(M, N, K, A, K, True, W, K, True, "gelu", bias=b, want_act=True)
    assert _rel(pre, pre_o.cpu().double()) < 2e-6 and _rel(post, post_o.cpu().double()) < 2e-6


def test_rows_kernel_limits(cuda):
    """Shapes the streaming kernel leaves to the tile kernel: few rows, K not a multiple of 32, an
    activation it has no instance of, outer-contiguous A -- the results do not depend on the choice."""
    from adell_mri_amd import _lib, ops

    g = torch.Generator().manual_seed(2)
    for M, N, K in ((1024, 384, 96), (65536, 96, 48)):
        A, W = torch.randn(M, K, generator=g).to(cuda), torch.randn(N, K, generator=g).to(cuda)
        a = ops.gemm_f16x3(M, N, K, A, K, True, W, K, True)
        with _lib.tuning(gemm_norows=1):
            b = ops.gemm_f16x3(M, N, K, A, K, True, W, K, True)
        assert torch.equal(a, b)
    A, W = torch.randn(65536, 96, generator=g).to(cuda), torch.randn(128, 96, generator=g).to(cuda)
    s1, _ = ops.gemm_f16x3_act(65536, 128, 96, A, 96, True, W, 96, True, "swish", want_act=True)
    with _lib.tuning(gemm_norows=1):
        s2, _ = ops.gemm_f16x3_act(65536, 128, 96, A, 96, True, W, 96, True, "swish", want_act=True)
    assert torch.equal(s1, s2)
