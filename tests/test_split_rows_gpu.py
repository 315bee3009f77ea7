"""Split rows (round 4): an ADN output whose only reader is a 3x3x3 stride-1 conv is stored as the
LDS row image of the f16x3 kernels ([hi c0-7 | hi c8-15 | lo c0-7 | lo c8-15] fp16 per voxel and
16-channel chunk, same bytes as fp32) and staged by copy. Checked: the converters, the fused
producer (adell_norm_act_fwd_split) against the fp32 producer, the forward and weight-gradient
kernels on rows against the same kernels on the fp32 tensor the rows stand for (interior, face and
ragged bricks; one source, two sources, mixed formats), and a residual block / small U-Net with
the format on and off."""
import itertools

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)


def test_converters_round_trip(cuda):
    from adell_mri_amd import ops

    g = torch.Generator().manual_seed(0)
    x = ops.ndhwc((torch.randn(2, 32, 6, 10, 12, generator=g) * 3.0).to(cuda))
    x[0, :, 0, 0, 0] = 0.0
    x[1, 3, 2, 2, 2] = 1e-6                       # far below the chunk's maximum
    rows, sr = ops.rows_from_f32(x, 9)            # |x| < ~16: 16 * 2^9 = 2^13
    assert rows.shape == x.shape and rows.dtype == torch.float32
    back = ops.rows_to_f32(rows, sr)
    assert float((back - x).abs().max()) <= 2.0 ** -22 * float(x.abs().max())
    small = float((back[1, 3, 2, 2, 2] - x[1, 3, 2, 2, 2]).abs())
    assert small <= 2.0 ** -25 / 2.0 ** 9         # the absolute floor of the lo half


@pytest.mark.parametrize("act,p", [("swish", 0.15), ("relu", 0.0), ("identity", 0.3), ("gelu", 0.0)])
@pytest.mark.parametrize("C,size", [(32, (16, 16, 16)), (64, (8, 12, 20)), (16, (24, 8, 8))])
def test_fused_producer_writes_the_rows_of_the_fp32_output(cuda, act, p, C, size):
    from adell_mri_amd import ops

    g = torch.Generator().manual_seed(C)
    x = ops.ndhwc((torch.randn(2, C, *size, generator=g) * 2.0 + 0.5).to(cuda))
    mean, rstd = ops.instance_stats(x)
    kw = dict(drop_p=p, seed=11, rng_offset=5, want_mask=True)
    ref, mask_ref = ops.norm_act_fwd(x, mean, rstd, act, **kw)
    rows, mask = ops.norm_act_fwd(x, mean, rstd, act, split_exp=6, **kw)
    sr = ops.SplitRows(6, ops.split_exponents(2, C, 6, x.device))
    got = ops.rows_to_f32(rows, sr)
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) <= 2.0 ** -21 * scale
    if p > 0:
        assert torch.equal(mask, mask_ref)


CASES = [  # N, C0, C1, Cout, size
    (2, 32, 0, 32, (16, 16, 16)),      # 8x8x4 bricks
    (2, 32, 0, 32, (64, 64, 64)),      # 8x8x8 bricks
    (1, 64, 0, 32, (76, 44, 68)),      # ragged bricks, four chunks
    (2, 64, 0, 64, (32, 32, 32)),      # 64-column tile
    (1, 32, 32, 64, (48, 48, 44)),     # virtual concat, ragged in z
    (2, 16, 0, 16, (32, 32, 32)),      # one chunk, half-empty column tile
    (1, 32, 0, 16, (24, 24, 24)),      # 16 x 16 weight-gradient tiles over a two-chunk rows source
]


def _operands(cuda, N, C0, C1, Cout, size, seed):
    from adell_mri_amd import ops

    g = torch.Generator().manual_seed(seed)
    x0 = ops.ndhwc((torch.randn(N, C0, *size, generator=g) * 1.5).to(cuda))
    x1 = ops.ndhwc(torch.randn(N, C1, *size, generator=g).to(cuda)) if C1 else None
    w = (torch.randn(Cout, C0 + C1, 3, 3, 3, generator=g) * 0.05).to(cuda)
    b = torch.randn(Cout, generator=g).to(cuda)
    return x0, x1, w, b


@pytest.mark.parametrize("N,C0,C1,Cout,size", CASES)
@pytest.mark.parametrize("which", ["both", "first", "second"])
def test_forward_on_rows_equals_forward_on_the_values_they_hold(cuda, N, C0, C1, Cout, size, which):
    from adell_mri_amd import ops

    if C1 == 0 and which != "both":
        pytest.skip("one source")
    if not ops.conv3d_rows_ok(N, size, C0, C1, Cout, 3, 1, 1):
        pytest.skip("this shape's launch plan does not stage rows")
    x0, x1, w, b = _operands(cuda, N, C0, C1, Cout, size, 3)
    wp = ops.pack_weight_f16x3(w, 0)
    r0, s0 = ops.rows_from_f32(x0, 9)
    v0 = ops.rows_to_f32(r0, s0)                 # the values the rows hold (22 bits of x0)
    r1 = s1 = v1 = None
    if x1 is not None:
        r1, s1 = ops.rows_from_f32(x1, 10)
        v1 = ops.rows_to_f32(r1, s1)
    y_ref, part_ref = ops.conv3d_fwd(v0, wp, b, Cout, 3, 1, 1, x1=v1, want_stats=True)
    use0, use1 = which in ("both", "first"), which in ("both", "second") and x1 is not None
    before = ops.ROWS_FALLBACKS[0]
    y, part = ops.conv3d_fwd(r0 if use0 else v0, wp, b, Cout, 3, 1, 1,
                             x1=(r1 if use1 else v1), want_stats=True,
                             rows0=s0 if use0 else None, rows1=s1 if use1 else None)
    assert ops.ROWS_FALLBACKS[0] == before, "the rows path was not taken"
    # same products, same accumulation order; only the power-of-two operand scale differs
    assert _rel(y, y_ref) <= 2e-6
    assert _rel(part, part_ref) <= 2e-5


@pytest.mark.parametrize("N,C0,C1,Cout,size", CASES)
def test_weight_gradient_on_rows(cuda, N, C0, C1, Cout, size):
    from adell_mri_amd import ops

    if not ops.conv3d_bwd_weight_rows_ok(N, size, C0, C1, Cout, 3, 1, 1):
        pytest.skip("not a z-ring problem")
    x0, x1, w, b = _operands(cuda, N, C0, C1, Cout, size, 5)
    g = torch.Generator().manual_seed(9)
    dy = ops.ndhwc((torch.randn(N, Cout, *size, generator=g) * 0.01).to(cuda))
    r0, s0 = ops.rows_from_f32(x0, 9)
    v0 = ops.rows_to_f32(r0, s0)
    r1 = s1 = v1 = None
    if x1 is not None:
        r1, s1 = ops.rows_from_f32(x1, 10)
        v1 = ops.rows_to_f32(r1, s1)
    dw_ref, db_ref = ops.conv3d_bwd_weight(v0, dy, 3, 1, 1, x1=v1, want_db=True, f16x3=True)
    before = ops.ROWS_FALLBACKS[0]
    dw, db = ops.conv3d_bwd_weight(r0, dy, 3, 1, 1, x1=r1, want_db=True, f16x3=True, rows0=s0,
                                   rows1=s1)
    assert ops.ROWS_FALLBACKS[0] == before
    assert _rel(dw, dw_ref) <= 5e-6
    assert torch.equal(db, db_ref)
    if x1 is not None:       # mixed: rows for the first source only
        dw2 = ops.conv3d_bwd_weight(r0, dy, 3, 1, 1, x1=v1, f16x3=True, rows0=s0)
        assert _rel(dw2, dw_ref) <= 5e-6


def test_a_plan_that_cannot_stage_rows_gets_the_values_back(cuda):
    """Stride 2 / tiny problems: the conv converts the rows (counted) and stays correct."""
    from adell_mri_amd import ops

    x0, _, w, b = _operands(cuda, 1, 32, 0, 32, (8, 8, 8), 2)
    wp = ops.pack_weight_f16x3(w, 0)
    r0, s0 = ops.rows_from_f32(x0, 9)
    v0 = ops.rows_to_f32(r0, s0)
    before = ops.ROWS_FALLBACKS[0]
    y, _ = ops.conv3d_fwd(r0, wp, b, 32, 3, 2, 1, rows0=s0)
    y_ref, _ = ops.conv3d_fwd(v0, wp, b, 32, 3, 2, 1)
    assert ops.ROWS_FALLBACKS[0] == before + 1
    assert torch.equal(y, y_ref)


def _block(cuda, c, p=0.15):
    from adell_mri_amd.modules.layers.adn_fn import get_adn_fn
    from adell_mri_amd.modules.layers.res_blocks import ResidualBlock3d

    torch.manual_seed(1)
    return ResidualBlock3d(c, 3, out_channels=c,
                           adn_fn=get_adn_fn(3, "instance", "swish", p)).to(cuda).train()


def _run_block(blk, x, r):
    from adell_mri_amd import functional as HF

    HF._dropout_counter = itertools.count(1)
    blk.zero_grad()
    xg = x.clone().requires_grad_(True)
    y = blk(xg)
    (y * r).sum().backward()
    torch.cuda.synchronize()
    return y.detach(), xg.grad.clone(), {k: v.grad.clone() for k, v in blk.named_parameters()}


def test_residual_block_with_and_without_rows(cuda, monkeypatch):
    """conv -> ADN -> conv (+ link) -> ADN: the inner ADN writes rows for the second conv. Outputs
    and gradients equal the fp32 format's to rounding of the operand scales."""
    from adell_mri_amd import functional as HF
    from adell_mri_amd import ops

    blk = _block(cuda, 32)
    x = torch.randn(2, 32, 32, 32, 32, device=cuda)
    r = torch.randn(2, 32, 32, 32, 32, device=cuda)
    made = []
    real = ops.norm_act_fwd
    monkeypatch.setattr(ops, "norm_act_fwd",
                        lambda *a, **k: made.append(k.get("split_exp")) or real(*a, **k))
    before = ops.ROWS_FALLBACKS[0]
    y_r, gx_r, gw_r = _run_block(blk, x, r)
    assert any(e is not None for e in made), "no site wrote rows"
    assert ops.ROWS_FALLBACKS[0] == before
    monkeypatch.setitem(HF.FLAGS, "no_rows", True)
    made.clear()
    y_f, gx_f, gw_f = _run_block(blk, x, r)
    assert all(e is None for e in made)
    assert _rel(y_r, y_f) <= 5e-6
    assert _rel(gx_r, gx_f) <= 2e-5
    scale = max(float(v.abs().max()) for v in gw_f.values())
    for k in gw_f:
        assert float((gw_r[k] - gw_f[k]).abs().max()) <= 5e-5 * scale, k


def _unet(cuda):
    from adell_mri_amd.modules.activations import activation_factory
    from adell_mri_amd.modules.segmentation.unet import UNet

    torch.manual_seed(2)
    return UNet(spatial_dimensions=3, conv_type="regular", link_type="residual",
                upscale_type="transpose", norm_type="instance", padding=1, dropout_param=0.0,
                activation_fn=activation_factory["swish"], in_channels=2, n_classes=2,
                depth=[16, 32, 32], kernel_sizes=[3] * 3, strides=[2] * 3).to(cuda).train()


@pytest.mark.parametrize("where", ["decoder_output", "link_output", "adn_output", "conv_pre_hook", "global"])
def test_a_forward_hook_never_sees_split_rows(cuda, monkeypatch, where):
    """A tensor in split-row format is fp32-typed memory holding fp16 row pairs: only the conv it
    was written for may read it. Any hook that could observe it -- on the ADN, on the link / decoder
    block around it, a pre-hook on the reading conv, a global module hook -- keeps the site in fp32
    (functional._hooked); hook-free sites keep their rows; logits agree with the hook-free run."""
    from adell_mri_amd import functional as HF
    from adell_mri_amd import ops
    from adell_mri_amd.modules.layers.adn_fn import ActDropNorm

    net = _unet(cuda)
    x = torch.rand(1, 2, 32, 32, 32, device=cuda)
    made = []
    real = ops.norm_act_fwd
    monkeypatch.setattr(ops, "norm_act_fwd",
                        lambda *a, **k: made.append(k.get("split_exp")) or real(*a, **k))
    ref = net(x, return_logits=True)[0].detach()
    plain_rows = sum(e is not None for e in made)
    assert plain_rows >= 3, made           # link -> decoder, decoder -> head and inner sites write rows
    seen = []

    def look(_m, _inp, out):
        seen.append(out)

    def look_pre(_m, inp):
        seen.extend(t for t in inp if torch.is_tensor(t))

    if where == "decoder_output":
        h = net.decoding_operations[-1].register_forward_hook(look)
    elif where == "link_output":
        hs = [l.register_forward_hook(look) for l in net.link_ops]
        h = type("H", (), {"remove": lambda self: [x.remove() for x in hs]})()
    elif where == "adn_output":
        adn = [m for m in net.decoding_operations[-1].modules() if isinstance(m, ActDropNorm)][-1]
        h = adn.register_forward_hook(look)
    elif where == "conv_pre_hook":
        from adell_mri_amd.modules.segmentation.unet import _first_conv
        h = _first_conv(net.final_layer).register_forward_pre_hook(look_pre)
    else:
        h = torch.nn.modules.module.register_module_forward_hook(look)
    try:
        made.clear()
        got = net(x, return_logits=True)[0].detach()
    finally:
        h.remove()
    assert seen, "the hook did not run"
    for t in seen:
        if torch.is_tensor(t):
            assert getattr(t, "_adell_rows", None) is None
    hooked_rows = sum(e is not None for e in made)
    # (whether a link block's last site writes rows depends on the decoder conv's launch plan at that
    # level: a hook there takes away at most what was there)
    assert hooked_rows <= plain_rows if where == "link_output" else hooked_rows < plain_rows
    if where == "global":
        assert hooked_rows == 0
    # the hooked tensor holds the values: feeding it to the head by hand reproduces the logits
    if where in ("decoder_output", "adn_output"):
        feat = [t for t in seen if torch.is_tensor(t)][-1]
        assert feat.shape[1] == 16 and torch.isfinite(feat).all() and float(feat.abs().max()) < 1e3
    assert _rel(got, ref) <= 5e-6
    # and once the hook is gone the rows are back
    made.clear()
    net(x, return_logits=True)
    assert sum(e is not None for e in made) == plain_rows
