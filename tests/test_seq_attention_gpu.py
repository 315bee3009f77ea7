"""Full-sequence attention on the packed projection output (functional._SeqAttnFn,
adell_attention_{fwd,bwd}_strided): the strided kernels against the contiguous entry points (same
code, other addresses: bit-identical, dropout included), the module path against the sliced form
it replaces and against torch's fp64 LayerNorm + scaled_dot_product_attention on the CPU
(linear_blocks.py:358-417)."""
import pytest
import torch
import torch.nn.functional as F

from adell_mri_amd import _lib, functional as HF, ops
from adell_mri_amd.modules.layers.linear_blocks import MultiHeadSelfAttention

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.mark.parametrize("B,H,T,A,Dv", [(2, 3, 216, 64, 64), (1, 2, 50, 32, 64), (3, 1, 130, 64, 32),
                                        (1, 4, 300, 128, 128)])
@pytest.mark.parametrize("drop", [0.0, 0.25])
def test_strided_kernels_are_the_contiguous_ones(cuda, B, H, T, A, Dv, drop):
    g = torch.Generator().manual_seed(T + A)
    per = 2 * A + Dv
    qkv = torch.randn(B * T, H * per, generator=g).to(cuda)
    bias = (torch.randn(H, T, T, generator=g) * 0.3).to(cuda)
    scale = A ** -0.5
    v5 = qkv.view(B, T, H, per).permute(0, 2, 1, 3)
    q = v5[..., :A].reshape(B * H, T, A).contiguous()
    k = v5[..., A:2 * A].reshape(B * H, T, A).contiguous()
    v = v5[..., 2 * A:].reshape(B * H, T, Dv).contiguous()
    out_c, lse_c = ops.attention_fwd(q, k, v, bias, scale, drop, 1234, 7)
    pak = (T * H * per, per, H * per)
    tok = (T * H * Dv, Dv, H * Dv)
    flat = qkv.view(-1)
    out_s = torch.full((B * T, H * Dv), float("nan"), device=cuda)
    lse_s = ops.attention_fwd_strided(flat, flat[A:], flat[2 * A:], out_s, pak * 3 + tok, bias, B, H,
                                      T, A, Dv, scale, drop, 1234, 7)
    want = out_c.view(B, H, T, Dv).transpose(1, 2).reshape(B * T, H * Dv)
    assert torch.equal(out_s, want) and torch.equal(lse_s, lse_c)

    dout = torch.randn(B * T, H * Dv, generator=g).to(cuda)
    dout_c = dout.view(B, T, H, Dv).transpose(1, 2).reshape(B * H, T, Dv).contiguous()
    dq, dk, dv = ops.attention_bwd(q, k, v, bias, out_c, dout_c, lse_c, scale, drop, 1234, 7)
    dqkv = torch.full_like(qkv, float("nan"))
    df = dqkv.view(-1)
    ops.attention_bwd_strided(flat, flat[A:], flat[2 * A:], out_s, dout, lse_s, df, df[A:],
                              df[2 * A:], pak * 3 + tok + tok + pak * 3, bias, B, H, T, A, Dv, scale,
                              drop, 1234, 7)
    d5 = dqkv.view(B, T, H, per).permute(0, 2, 1, 3)
    assert torch.equal(d5[..., :A].reshape(B * H, T, A), dq)
    assert torch.equal(d5[..., A:2 * A].reshape(B * H, T, A), dk)
    assert torch.equal(d5[..., 2 * A:].reshape(B * H, T, Dv), dv)


def test_bad_strides_are_refused(cuda):
    x = torch.zeros(4 * 64 * 192, device=cuda)
    out = torch.zeros(4 * 64 * 64, device=cuda)
    ok = (64 * 192, 0, 192)
    with pytest.raises(_lib.AdellHipError):       # rows closer than the head size
        ops.attention_fwd_strided(x, x[64:], x[128:], out, (64 * 192, 0, 32) + ok + ok + (64 * 64, 0, 64),
                                  None, 4, 1, 64, 64, 64, 0.125)
    with pytest.raises(_lib.AdellHipError):       # stride not a multiple of four elements
        ops.attention_fwd_strided(x, x[64:], x[128:], out, (64 * 192, 0, 194) + ok + ok + (64 * 64, 0, 64),
                                  None, 4, 1, 64, 64, 64, 0.125)
    with pytest.raises(_lib.AdellHipError):       # unaligned pointer
        ops.attention_fwd_strided(x[1:], x[64:], x[128:], out, ok * 3 + (64 * 64, 0, 64), None, 4, 1,
                                  64, 64, 64, 0.125)
    with pytest.raises(_lib.AdellHipError):       # no MFMA instance for 48-wide heads
        ops.attention_fwd_strided(x, x[48:], x[96:], out, (64 * 144, 0, 144) * 3 + (64 * 48, 0, 48),
                                  None, 4, 1, 64, 48, 48, 0.125)
    assert not HF.seq_attention_ok(100, 48, 48) and HF.seq_attention_ok(100, 64, 64)


@pytest.mark.parametrize("B,T,heads,att,hid,masked", [(2, 216, 3, 192, 192, False),
                                                      (1, 100, 2, 64, 128, True),
                                                      (1, 100, 2, 64, 128, False),
                                                      (2, 72, 2, 256, 64, True)])
def test_module_against_the_sliced_form_and_fp64(cuda, B, T, heads, att, hid, masked):
    torch.manual_seed(T)
    dim = 96
    m = MultiHeadSelfAttention(dim, att, hid, dim, n_heads=heads).to(cuda)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "norm" in n:
                p.add_(torch.randn_like(p) * 0.2)
    x = torch.randn(B, T, dim, device=cuda, requires_grad=True)
    mask = (torch.randn(B, T, T, device=cuda) * 0.5) if masked else None
    gy = torch.randn(B, T, dim, device=cuda)
    a, h = att // heads, hid // heads
    assert HF.seq_attention_ok(T, a, h)

    def run(flag):
        HF.FLAGS["no_seq_attention"] = flag
        try:
            for p in list(m.parameters()) + [x]:
                p.grad = None
            y = m(x, mask=mask)
            y.backward(gy)
            return [y.detach().clone()] + [p.grad.clone() for p in [x] + list(m.parameters())]
        finally:
            HF.FLAGS["no_seq_attention"] = False

    new, old = run(False), run(True)
    names = ["y", "x"] + [n for n, _ in m.named_parameters()]

    # torch fp64 on the CPU: the reference module's arithmetic
    md = {n: p.detach().cpu().double().requires_grad_(True) for n, p in m.named_parameters()}
    xd = x.detach().cpu().double().requires_grad_(True)
    qkv = (xd @ md["qkv.weight"].T).reshape(B, T, heads, 2 * a + h).permute(0, 2, 1, 3)
    q = F.layer_norm(qkv[..., :a], (a,), md["q_norm.weight"], md["q_norm.bias"], m.q_norm.eps)
    k = F.layer_norm(qkv[..., a:2 * a], (a,), md["k_norm.weight"], md["k_norm.bias"], m.k_norm.eps)
    am = None if mask is None else mask.cpu().double().unsqueeze(1)
    o = F.scaled_dot_product_attention(q, k, qkv[..., 2 * a:], attn_mask=am)
    o = o.transpose(1, 2).reshape(B, T, hid)
    y = o @ md["output_layer.weight"].T + md["output_layer.bias"]
    y.backward(gy.cpu().double())
    ref = [y.detach(), xd.grad] + [md[n].grad for n, _ in m.named_parameters()]

    # a constant added to every key moves all the scores of a query together: the true gradient of
    # k_norm.bias is zero and what the kernels return is rounding noise -- measured against the
    # size of q_norm.bias' gradient instead of its own
    floor = float(ref[names.index("q_norm.bias")].abs().max())
    bad = {}
    for n, u, v, r in zip(names, new, old, ref):
        scale = floor if n == "k_norm.bias" else float(r.abs().max())
        e_new = float((u.cpu().double() - r).abs().max()) / scale
        e_old = float((v.cpu().double() - r).abs().max()) / scale
        if e_new > 5e-5 or e_old > 5e-5:
            bad[n] = (e_new, e_old)
    assert not bad, f"(new, sliced) errors against fp64: {bad}"
