"""LayerNorm over rows (csrc/window.hip) against torch, over row lengths, strided sources (the q / k
slices of a packed projection, two-level row offsets) and strided gradient destinations."""
import pytest
import torch
import torch.nn.functional as F

from adell_mri_amd import ops

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("C", [4, 8, 12, 20, 24, 36, 48, 64, 96, 100, 192, 384, 512])
@pytest.mark.parametrize("layout", ["dense", "heads", "two_level"])
def test_rows_and_strides_against_torch(cuda, C, layout):
    g = torch.Generator().manual_seed(C)
    rows = 777
    if layout == "dense":
        src = torch.randn(rows, C, generator=g).to(cuda)
        view, inner, so, si = src, 1, C, 0
        x_ref = src
    elif layout == "heads":          # rows = (token, head): row r at r * per inside [tokens, H * per]
        per = 2 * C + 8
        rows = 259 * 3
        src = torch.randn(259, 3 * per, generator=g).to(cuda)
        view, inner, so, si = src.view(-1)[C:], 1, per, 0          # the "k" slice
        x_ref = src.view(259, 3, per)[:, :, C:2 * C].reshape(rows, C)
    else:                            # (r // inner) * so + (r % inner) * si
        per, H = C + 4, 3
        rows = 259 * H
        src = torch.randn(259, H * per + 8, generator=g).to(cuda)
        view, inner, so, si = src.view(-1), H, H * per + 8, per
        x_ref = src[:, :H * per].reshape(259, H, per)[:, :, :C].reshape(rows, C)
    gamma = (1 + 0.3 * torch.randn(C, generator=g)).to(cuda)
    beta = torch.randn(C, generator=g).to(cuda)
    dy = torch.randn(rows, C, generator=g).to(cuda)

    def run():
        y, m, r = ops.layernorm_rows_fwd(view, rows, C, inner, so, si, gamma, beta, 1e-5)
        dxbuf = torch.zeros_like(src)
        dview = dxbuf.view(-1)[C:] if layout == "heads" else dxbuf.view(-1)
        dg, db = ops.layernorm_rows_bwd(view, dy, gamma, m, r, rows, C, inner, so, si, dview,
                                        so, si, True)
        return y, dxbuf, dg, db

    yv, dxv, dgv, dbv = run()
    ys, dxs, dgs, dbs = run()
    assert torch.equal(yv, ys) and torch.equal(dxv, dxs) and torch.equal(dgv, dgs)   # deterministic
    xr = x_ref.detach().clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    yt = F.layer_norm(xr, (C,), gr, br, 1e-5)
    yt.backward(dy)
    if layout == "dense":
        dx_ref = xr.grad
        pick = lambda t: t
    elif layout == "heads":
        pick = lambda t: t.view(259, 3, 2 * C + 8)[:, :, C:2 * C].reshape(rows, C)
        dx_ref = xr.grad
    else:
        pick = lambda t: t[:, :3 * (C + 4)].reshape(259, 3, C + 4)[:, :, :C].reshape(rows, C)
        dx_ref = xr.grad
    for got_y, got_dx, got_dg, got_db in ((yv, dxv, dgv, dbv),):
        assert torch.allclose(got_y, yt, rtol=1e-5, atol=1e-5)
        assert torch.allclose(pick(got_dx), dx_ref, rtol=1e-4, atol=1e-5)
        assert torch.allclose(got_dg, gr.grad, rtol=1e-4, atol=1e-4)
        assert torch.allclose(got_db, br.grad, rtol=1e-4, atol=1e-4)
    # nothing written outside the slices
    mask = torch.ones_like(dxv, dtype=torch.bool)
    if layout == "dense":
        mask[:] = False
    elif layout == "heads":
        mask.view(259, 3, 2 * C + 8)[:, :, C:2 * C] = False
    else:
        mask[:, :3 * (C + 4)].unflatten(1, (3, C + 4))[:, :, :C] = False
    assert not bool(mask.any()) or float(dxv[mask].abs().max()) == 0.0


@pytest.mark.parametrize("rows,C,inner,so,si", [(256, 32, 1, 32, 0), (256, 16, 1, 16, 0), (1024, 4, 4, 48, 12),
                                               (1024, 8, 4, 96, 24), (2048, 4, 1, 4, 0), (128, 64, 1, 64, 0),
                                               (512, 16, 4, 192, 48), (512, 16, 1, 16, 0), (4096, 2, 1, 2, 0),
                                               (30, 8, 1, 8, 0), (5, 512, 1, 512, 0)])
def test_the_row_shapes_of_the_small_swin_fixtures(cuda, rows, C, inner, so, si):
    g = torch.Generator().manual_seed(rows + C)
    outer = (rows + inner - 1) // inner
    src = torch.randn(outer * so + 64, generator=g).to(cuda)
    idx = (torch.arange(rows) // inner) * so + (torch.arange(rows) % inner) * si
    x_ref = torch.stack([src[i:i + C] for i in idx.tolist()])
    gamma = (1 + 0.3 * torch.randn(C, generator=g)).to(cuda)
    beta = torch.randn(C, generator=g).to(cuda)
    want = F.layer_norm(x_ref, (C,), gamma, beta, 1e-5)
    for off in (0, C):        # the q and the k slice of a packed row
        if off and si == 0 and so == C:
            continue
        x_off = torch.stack([src[i + off:i + off + C] for i in idx.tolist()])
        y, m, r = ops.layernorm_rows_fwd(src[off:], rows, C, inner, so, si, gamma, beta, 1e-5)
        assert torch.allclose(y, F.layer_norm(x_off, (C,), gamma, beta, 1e-5), rtol=1e-5, atol=1e-5), off
        assert torch.allclose(m, x_off.mean(1), rtol=1e-5, atol=1e-6)
    assert want.shape == (rows, C)
