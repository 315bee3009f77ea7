"""Backward of a norm -> dropout -> activation site fused into the backward-data kernel of the conv
that reads it (adn_fn.py:140-152 feeding unet.py:260-273 / res_blocks.py:150-178): the epilogue of
adell_conv3d_bwd_data_f16x3_adn stores dt = dout * act'(u) * keep / (1 - p) and the two per-channel
sums, adell_norm_act_bwd_from_dt finishes the site. Checked against the unfused kernels (backward-
data, then the two-pass site backward with the regenerated Philox mask) and, through a module, against
torch autograd on the CPU."""
import itertools

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)


class _Site:
    def __init__(self, x, mean, rstd, mask, drop_p, act, act_p=0.0):
        self.x, self.mean, self.rstd, self.mask = x, mean, rstd, mask
        self.drop_p, self.act, self.act_p = drop_p, act, act_p


def _site(cuda, g, n, c, size, act, p, seed):
    from adell_mri_amd import ops

    y = ops.ndhwc((torch.randn(n, c, *size, generator=g) * 1.7 + 0.3).to(cuda))
    mean, rstd = ops.instance_stats(y)
    out, mask = ops.norm_act_fwd(y, mean, rstd, act, act_p=0.2 if act == "leaky_relu" else 0.0,
                                 drop_p=p, seed=seed, rng_offset=7, want_mask=True)
    return _Site(y, mean, rstd, mask, p, act, 0.2 if act == "leaky_relu" else 0.0), out


@pytest.mark.parametrize("n,c0,c1,cout,size", [(2, 32, 0, 32, (16, 16, 16)),     # 8x8x8 bricks
                                               (2, 32, 0, 48, (32, 32, 32)),     # 8x8x4 bricks
                                               (1, 64, 0, 32, (64, 64, 32)),     # 64-column tile
                                               (1, 32, 32, 64, (32, 64, 64))])   # two destinations
@pytest.mark.parametrize("act,p", [("swish", 0.15), ("relu", 0.0), ("leaky_relu", 0.3),
                                   ("identity", 0.5)])
def test_fused_epilogue_equals_the_two_pass_backward(cuda, n, c0, c1, cout, size, act, p):
    from adell_mri_amd import functional as HF
    from adell_mri_amd import ops

    g = torch.Generator().manual_seed(c0 + 3 * c1 + cout)
    k, st, pad = (3, 3, 3), (1, 1, 1), (1, 1, 1)
    nt = ops.conv3d_bwd_data_adn_ntiles(size, n, c0, c1, cout, k, st, pad)
    assert nt > 0, "this shape must take the fused epilogue"
    w = (torch.randn(cout, c0 + c1, 3, 3, 3, generator=g) * 0.05).to(cuda)
    wpb = HF._packed(w, 1)
    dy = ops.ndhwc(torch.randn(n, cout, *size, generator=g).to(cuda))
    s0, _ = _site(cuda, g, n, c0, size, act, p, 11)
    s1 = _site(cuda, g, n, c1, size, act, p, 12)[0] if c1 else None
    add0 = ops.ndhwc(torch.randn(n, c0, *size, generator=g).to(cuda)) if c1 == 0 else None

    for first_only in ((False, True) if c1 else (False,)):
        sites = (None, s1) if first_only else (s0, s1)       # a plain first destination as well
        dt0, dt1, part = ops.conv3d_bwd_data_adn(dy, wpb, size, c0, c1, k, st, pad, nt,
                                                 site0=sites[0], site1=sites[1], add0=add0)
        da0, da1 = ops.conv3d_bwd_data(dy, wpb, size, c0, c1, k, st, pad)
        if add0 is not None:
            da0 = da0 + add0
        for site, dt, da, poff in ((sites[0], dt0, da0, 0), (sites[1], dt1, da1, c0)):
            if site is None:
                if da is not None:
                    assert _rel(dt, da) <= 2e-6       # plain destination: the gradient itself
                continue
            want, _, _ = ops.norm_act_bwd(site.x, da, site.mean, site.rstd, site.act,
                                          act_p=site.act_p, drop_p=p,
                                          seed=11 if site is s0 else 12, rng_offset=7)
            got = ops.norm_act_bwd_from_dt(site.x, dt, site.mean, site.rstd, part, poff)
            assert _rel(got, want) <= 2e-5, (act, p, poff)


def test_keep_bits_are_the_forward_mask(cuda):
    from adell_mri_amd import ops

    g = torch.Generator().manual_seed(5)
    x = ops.ndhwc((torch.rand(2, 32, 8, 8, 12, generator=g) + 0.5).to(cuda))     # never zero
    out, mask = ops.norm_act_fwd(x, None, None, "identity", drop_p=0.3, seed=3, rng_offset=2,
                                 want_mask=True)
    kept = (out.permute(0, 2, 3, 4, 1).reshape(2, -1) != 0).cpu()               # [N, V * C]
    words = mask.cpu().view(2, -1, 4)                                           # [N, groups, 4]
    el = torch.arange(kept.shape[1])
    bits = (words[:, el >> 8, el & 3] >> ((el >> 2) & 63)) & 1
    assert torch.equal(bits.bool(), kept)
    assert 0.25 < 1.0 - kept.float().mean().item() < 0.35


def _block(cuda, c, act="swish", p=0.15):
    from adell_mri_amd.modules.layers.adn_fn import get_adn_fn
    from adell_mri_amd.modules.layers.res_blocks import ResidualBlock3d

    torch.manual_seed(1)
    return ResidualBlock3d(c, 3, out_channels=c,
                           adn_fn=get_adn_fn(3, "instance", act, p)).to(cuda).train()


def _run_block(blk, x, r):
    from adell_mri_amd import functional as HF

    HF._dropout_counter = itertools.count(1)       # the same dropout masks in both runs
    blk.zero_grad()
    xg = x.clone().requires_grad_(True)
    y = blk(xg)
    (y * r).sum().backward()
    return y.detach(), xg.grad.clone(), {k: v.grad.clone() for k, v in blk.named_parameters()}


def test_residual_block_gradients_do_not_depend_on_the_fusion(cuda, monkeypatch):
    """conv -> ADN -> conv (+ link) -> ADN with dropout: the inner site hands half of its backward
    to the second conv's backward-data kernel; same gradients as with the switch off."""
    from adell_mri_amd import functional as HF
    from adell_mri_amd import ops

    blk = _block(cuda, 32)
    x = torch.randn(2, 32, 16, 16, 16, device=cuda)
    r = torch.randn(2, 32, 16, 16, 16, device=cuda)
    calls = []
    real = ops.conv3d_bwd_data_adn
    monkeypatch.setattr(ops, "conv3d_bwd_data_adn", lambda *a, **k: calls.append(1) or real(*a, **k))
    y_f, gx_f, gw_f = _run_block(blk, x, r)
    assert calls, "the fused epilogue did not run"
    monkeypatch.setitem(HF.FLAGS, "no_adn_fuse", True)
    y_p, gx_p, gw_p = _run_block(blk, x, r)
    assert len(calls) == 1
    assert torch.equal(y_f, y_p)
    assert _rel(gx_f, gx_p) <= 2e-5
    scale = max(float(v.abs().max()) for v in gw_p.values())
    for k in gw_p:
        if k.endswith("bias"):
            # a bias in front of an instance norm has gradient zero: both runs hold rounding noise
            assert float((gw_f[k] - gw_p[k]).abs().max()) <= 1e-5 * scale, k
        else:
            assert _rel(gw_f[k], gw_p[k]) <= 5e-5, k


def test_launch_plan_flipped_between_forward_and_backward(cuda):
    """adell_set_tuning between the forward and the backward of a block whose inner site rides the
    second conv's backward-data epilogue: the partial-sum rows are the CURRENT plan's (the forward
    planned 8x8x8 bricks, the backward runs 8x8x4 ones: twice the rows), never a count cached from
    the forward -- the gradients equal the unflipped run, and the C entry refuses a stale count
    outright (round 3's A/B fault; include/adell_hip.h, ABI version 2). Run once."""
    from adell_mri_amd import _lib
    from adell_mri_amd import functional as HF
    from adell_mri_amd import ops

    blk = _block(cuda, 32)
    x = torch.randn(2, 32, 64, 64, 64, device=cuda)
    r = torch.randn(2, 32, 64, 64, 64, device=cuda)
    y0, gx0, gw0 = _run_block(blk, x, r)

    HF._dropout_counter = itertools.count(1)
    blk.zero_grad()
    xg = x.clone().requires_grad_(True)
    y = blk(xg)                                   # planned under the default switches
    size, k, st, pad = (64, 64, 64), (3, 3, 3), (1, 1, 1), (1, 1, 1)
    rows_fwd = ops.conv3d_bwd_data_adn_ntiles(size, 2, 32, 0, 32, k, st, pad)
    with _lib.tuning(igemm_no8=1):
        rows_bwd = ops.conv3d_bwd_data_adn_ntiles(size, 2, 32, 0, 32, k, st, pad)
        assert rows_bwd not in (0, rows_fwd)
        # the stale count is refused before anything is launched
        w = blk.op[2].weight if hasattr(blk, "op") else next(blk.parameters())
        dy = ops.ndhwc(torch.randn(2, w.shape[0], *size, device=cuda))
        site = _site(cuda, torch.Generator().manual_seed(3), 2, 32, size, "swish", 0.0, 5)[0]
        with pytest.raises(_lib.AdellHipError, match="rows"):
            ops.conv3d_bwd_data_adn(dy, HF._packed(w, 1), size, 32, 0, k, st, pad, rows_fwd,
                                    site0=site)
        (y * r).sum().backward()
    torch.cuda.synchronize()
    assert torch.equal(y.detach(), y0)
    assert _rel(xg.grad, gx0) <= 2e-5
    scale = max(float(v.abs().max()) for v in gw0.values())
    for name, v in blk.named_parameters():
        assert float((v.grad - gw0[name]).abs().max()) <= 5e-5 * scale, name


def test_residual_block_matches_torch_autograd(cuda, monkeypatch):
    """The fused path against stock torch (fp64, CPU; no dropout: torch's mask stream differs)."""
    from adell_mri_amd import ops

    blk = _block(cuda, 32, act="swish", p=0.0)
    x = torch.randn(2, 32, 16, 16, 16, device=cuda)
    r = torch.randn(2, 32, 16, 16, 16, device=cuda)
    calls = []
    real = ops.conv3d_bwd_data_adn
    monkeypatch.setattr(ops, "conv3d_bwd_data_adn", lambda *a, **k: calls.append(1) or real(*a, **k))
    _, gx, gw = _run_block(blk, x, r)
    assert calls, "the fused epilogue did not run"
    sd = {k: v.detach().cpu().double() for k, v in blk.state_dict().items()}
    xc = x.cpu().double().requires_grad_(True)
    ws = [sd[k].requires_grad_(True) for k in ("op.0.weight", "op.2.weight")]
    F = torch.nn.functional
    h = F.conv3d(xc, ws[0], sd["op.0.bias"], padding=1)
    h = F.silu(F.instance_norm(h, eps=1e-5))
    h = F.conv3d(h, ws[1], sd["op.2.bias"], padding=1) + xc
    out = F.silu(F.instance_norm(h, eps=1e-5))
    (out * r.cpu().double()).sum().backward()
    assert _rel(gx.cpu().double(), xc.grad) <= 1e-4
    assert _rel(gw["op.0.weight"].cpu().double(), ws[0].grad) <= 1e-4
    assert _rel(gw["op.2.weight"].cpu().double(), ws[1].grad) <= 1e-4


@pytest.mark.parametrize("n,c,co,size", [(2, 32, 1, (16, 16, 24)), (1, 16, 3, (8, 12, 20)),
                                         (2, 64, 4, (8, 8, 8)), (1, 32, 2, (40, 24, 8))])
@pytest.mark.parametrize("act,p", [("swish", 0.1), ("relu", 0.0), ("leaky_relu", 0.25)])
def test_lowrank_site_backward_equals_conv_backward_data_then_site(cuda, n, c, co, size, act, p):
    """The site in front of the 1x1x1 logits conv (unet.py:626-655): its backward from (dy, w)
    against the 1x1x1 backward-data kernel followed by the ordinary site backward."""
    from adell_mri_amd import ops

    g = torch.Generator().manual_seed(c + 7 * co)
    site, _ = _site(cuda, g, n, c, size, act, p, 21)
    w = (torch.randn(co, c, 1, 1, 1, generator=g) * 0.3).to(cuda)
    dy = ops.ndhwc(torch.randn(n, co, *size, generator=g).to(cuda))
    assert ops.norm_act_lowrank_ok(site.x, co)
    da, _ = ops.conv1_small_bwd_data(dy, w, size, c, 0)
    want, _, _ = ops.norm_act_bwd(site.x, da, site.mean, site.rstd, act, act_p=site.act_p,
                                  drop_p=p, seed=21, rng_offset=7)
    got = ops.norm_act_bwd_lowrank(site.x, dy, w, site.mean, site.rstd, act, act_p=site.act_p,
                                   drop_p=p, seed=21, rng_offset=7)
    assert _rel(got, want) <= 2e-6
    # without a normalisation in the site: the elementwise pass alone
    want, _, _ = ops.norm_act_bwd(site.x, da, None, None, act, act_p=site.act_p, drop_p=p,
                                  seed=21, rng_offset=7)
    got = ops.norm_act_bwd_lowrank(site.x, dy, w, None, None, act, act_p=site.act_p, drop_p=p,
                                   seed=21, rng_offset=7)
    assert _rel(got, want) <= 2e-6


def test_lowrank_rejects_what_the_kernels_do_not_cover(cuda):
    from adell_mri_amd import ops
    from adell_mri_amd._lib import AdellHipError

    x = ops.ndhwc(torch.randn(1, 24, 4, 4, 4).to(cuda))        # 24 channels: not a power of two
    mean, rstd = ops.instance_stats(x)
    assert not ops.norm_act_lowrank_ok(x, 1)
    with pytest.raises(AdellHipError):
        ops.norm_act_bwd_lowrank(x, ops.ndhwc(torch.randn(1, 1, 4, 4, 4).to(cuda)),
                                 torch.randn(1, 24, 1, 1, 1).to(cuda), mean, rstd, "swish")
    x = ops.ndhwc(torch.randn(1, 32, 4, 4, 4).to(cuda))
    assert not ops.norm_act_lowrank_ok(x, 5)


def test_head_gradients_do_not_depend_on_the_lowrank_site(cuda, monkeypatch):
    """conv -> ADN -> conv 1x1x1 (C -> 1) -> sigmoid as the U-Net builds its final layer: the same
    parameter and input gradients with the switch off, and the 1x1x1 backward-data kernel does not
    run with it on."""
    from adell_mri_amd import functional as HF
    from adell_mri_amd import ops
    from adell_mri_amd.modules.segmentation.unet import UNet

    torch.manual_seed(3)
    from adell_mri_amd.modules.layers.adn_fn import activation_factory
    net = UNet(spatial_dimensions=3, upscale_type="transpose", norm_type="instance", padding=1,
               dropout_param=0.1, activation_fn=activation_factory["swish"], in_channels=2,
               n_classes=2, depth=[16, 32], kernel_sizes=[3, 3], strides=[2, 2]).to(cuda)
    x = torch.randn(2, 2, 16, 16, 16, device=cuda)
    t = (torch.rand(2, 1, 16, 16, 16, device=cuda) > 0.5).float()
    calls = []
    real = ops.conv1_small_bwd_data
    monkeypatch.setattr(ops, "conv1_small_bwd_data", lambda *a, **k: calls.append(1) or real(*a, **k))

    def run():
        net.zero_grad(set_to_none=True)
        xx = x.clone().requires_grad_(True)
        out = net(xx)
        prob = out[0] if isinstance(out, tuple) else out
        loss = ((prob - t) ** 2).mean()
        loss.backward()
        return prob.detach(), xx.grad.clone(), {k: p.grad.clone() for k, p in net.named_parameters()
                                                if p.grad is not None}

    net.train()
    import itertools as it
    monkeypatch.setattr(HF, "_dropout_counter", it.count(100))
    y_f, gx_f, gw_f = run()
    assert not calls, "the 1x1x1 backward-data kernel ran although the site takes (dy, w)"
    monkeypatch.setitem(HF.FLAGS, "no_adn_fuse", True)
    monkeypatch.setattr(HF, "_dropout_counter", it.count(100))
    y_p, gx_p, gw_p = run()
    assert calls
    assert torch.equal(y_f, y_p)
    scale = max(float(v.abs().max()) for v in gw_p.values())
    assert _rel(gx_f, gx_p) <= 5e-5
    for k in gw_p:
        assert float((gw_f[k] - gw_p[k]).abs().max()) <= 5e-5 * scale, k


def test_a_hook_on_a_site_output_sees_the_true_gradient(cuda):
    """A tensor hook (or retain_grad) on an ADN output that a fused site epilogue or the low-rank head
    path would otherwise serve with dt / a placeholder: the site then keeps its own backward and the
    watcher sees d loss / d output."""
    from adell_mri_amd import functional as HF
    from adell_mri_amd.modules.layers.conv import Conv3d

    torch.manual_seed(0)
    conv = Conv3d(32, 32, 3, padding=1).to(cuda)
    x = torch.randn(2, 32, 16, 16, 16, device=cuda, requires_grad=True)
    r = torch.randn(2, 32, 16, 16, 16, device=cuda)

    def run(watch):
        HF._dropout_counter = itertools.count(1)
        conv.zero_grad()
        x.grad = None
        h = HF.norm_drop_act(x, norm="instance", act="swish", drop_p=0.0, training=True)
        seen = []
        if watch:
            h.register_hook(lambda g: seen.append(g.detach().clone()))
        h = HF.single_use(h)
        (conv(h) * r).sum().backward()
        return x.grad.clone(), seen

    gx_plain, _ = run(False)
    gx_watch, seen = run(True)
    assert len(seen) == 1
    assert _rel(gx_watch, gx_plain) <= 2e-5
    # the watcher's tensor is the gradient with respect to the ADN OUTPUT: conv backward-data of r
    want = torch.nn.grad.conv3d_input(x.shape, conv.weight.detach().cpu(), r.cpu(), padding=1)
    assert _rel(seen[0].cpu(), want) <= 1e-4
