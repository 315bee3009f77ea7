"""The benchmark-size path pinned to the REAL reference (-m gpu).

``tests/golden/unet3d_cfg2_full*.npz`` were written by ``oracle/make_golden.py full`` from the
imported reference (``UNet.forward`` unet.py:751-843 + dice / focal + backward) at 1x2x128^3 and
1x2x256x256x128 on seeded inputs that are regenerated here (``oracle/fullsize.py``; the fixture
carries an input checksum). Stored: logit statistics, 64 sampled voxels, a line and the eight
corners, the loss terms, and for every parameter gradient its L2 norm, absolute maximum and 16
sampled entries.

Bars: logits 1e-4 of the logit range (north_star), loss 1e-4 relative, gradient norms 1e-3
relative, sampled gradient entries 2e-3 of the gradient's absolute maximum.

Also here: the specialised f16x3 instances (SPEC = 1: 64-channel tile, 8x8x4 brick; SPEC = 3:
32-channel tile, 8x8x8 brick) and the z-ring weight-gradient kernel directly against the fp64
C oracle on a shape with interior AND ragged bricks, and f16x3 against fp32-MFMA at 2x128^3."""
import os

import numpy as np
import pytest
import torch

from adell_mri_amd import functional as HF
from adell_mri_amd import _lib, ops
from adell_mri_amd.modules.activations import activation_factory
from adell_mri_amd.modules.segmentation.unet import UNet
from oracle import cops
from oracle.fullsize import full_inputs, sample_positions, zlib_crc
from oracle.torch_ref.unet import compound_loss, dice_loss, focal_loss
from oracle.weights import tensor_for

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CFG2 = dict(spatial_dimensions=3, conv_type="regular", link_type="residual",
            upscale_type="transpose", norm_type="instance", padding=1, dropout_param=0.15,
            activation_fn="swish", in_channels=2, n_classes=2, depth=[32, 32, 64, 128, 256],
            kernel_sizes=[3] * 5, strides=[2] * 5)


def build(device):
    kw = dict(CFG2)
    kw["activation_fn"] = activation_factory[kw["activation_fn"]]
    net = UNet(**kw)
    net.load_state_dict({k: torch.from_numpy(tensor_for(k, v.shape))
                         for k, v in net.state_dict().items()})
    return net.to(device).eval()


@pytest.fixture
def precision(request):
    old = HF.CONV_PRECISION
    HF.set_conv_precision(request.param)
    yield request.param
    HF.set_conv_precision(old)


def _grad_scale(g, k, ref_max, top):
    """Scale a gradient error is measured against (as tests/cases.py::grad_rel_err): the
    gradient's own maximum, floored by 10 % of the sibling weight's for a bias (a bias in front
    of an instance norm has a mathematically zero gradient: rounding noise in the reference
    too) and by 5e-4 of the largest gradient of the network."""
    scale = max(ref_max, 5e-4 * top)
    if k.endswith(".bias") and ("gnorm:" + k[:-5] + ".weight") in g.files:
        scale = max(scale, 1e-1 * float(g["gnorm:" + k[:-5] + ".weight"][1]))
    return scale


def _check_forward(g, logits, tag):
    lg = logits.detach().float().cpu()
    stats = g["logit_stats"]
    rng = float(stats[3] - stats[2])
    flat = lg.reshape(-1)
    got_stats = [float(lg.double().mean()), float(lg.double().std()), float(lg.min()),
                 float(lg.max()), float(lg.double().abs().mean())]
    err = {
        "sampled": float(np.abs(flat[g["logit_pos"]].numpy() - g["logit_val"]).max()) / rng,
        "line": float(np.abs(lg[0, 0, lg.shape[2] // 2, lg.shape[3] // 2, :].numpy()
                             - g["logit_line"]).max()) / rng,
        "corners": float(np.abs(lg[0, 0][::lg.shape[2] - 1, ::lg.shape[3] - 1,
                                         ::lg.shape[4] - 1].numpy() - g["logit_corners"]).max()) / rng,
        "stats": float(np.abs(np.array(got_stats) - stats).max()) / rng,
    }
    print(f"{tag}: full-size logits vs reference, error / logit range: {err}")
    assert max(err.values()) < 1e-4, err
    return err


@pytest.mark.parametrize("precision", ["f16x3", "fp32"], indirect=True)
def test_cfg2_128_cubed_matches_reference(cuda, precision):
    g = np.load(os.path.join(GOLD, "unet3d_cfg2_full.npz"))
    shape = tuple(int(v) for v in g["shape"])
    x, y = full_inputs(shape, int(g["seed"]))
    assert abs(float(x.double().sum()) - g["x_checksum"][0]) < 1e-6 * g["x_checksum"][0]
    assert float(y.double().sum()) == g["x_checksum"][1]
    net = build(cuda)
    logits = net(x.to(cuda), return_logits=True)[0]
    _check_forward(g, logits, f"cfg2 128^3 [{precision}]")
    prob = torch.sigmoid(logits)
    yd = y.to(cuda)
    d, f = dice_loss(prob, yd), focal_loss(prob, yd)
    np.testing.assert_allclose(d.detach().cpu().numpy(), g["dice"], rtol=1e-4)
    np.testing.assert_allclose(f.detach().cpu().numpy(), g["focal"], rtol=1e-4)
    del prob, d, f
    # the product's own probability head + fused loss, then every parameter gradient
    prob = net(x.to(cuda))[0]
    loss = compound_loss(prob, yd)
    np.testing.assert_allclose(float(loss.detach()), float(g["loss"]), rtol=1e-4)
    loss.backward()
    # The yardstick is the reference network in DOUBLE precision (unet3d_cfg2_full_fp64.npz, same
    # inputs / weights / sampled positions: `python oracle/make_golden.py full64`). The reference's
    # own fp32 gradients deviate from it by its summation noise over 2 M voxels per channel; the HIP
    # gradients must be as close to fp64 as the reference's fp32 ones are (x 3), or within 1e-4.
    g64 = np.load(os.path.join(GOLD, "unet3d_cfg2_full_fp64.npz"))
    worst = {"hip_norm": (0.0, ""), "ref_norm": (0.0, ""), "hip_entry": (0.0, ""),
             "ref_entry": (0.0, "")}
    top = max(float(g64["gnorm:" + str(k)][1]) for k in g64["grad_keys"])
    rows = []
    for k, p in net.named_parameters():
        if ("gnorm:" + k) not in g.files:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        gr = p.grad.detach().float().cpu().reshape(-1)
        ref_norm, ref_max = (float(v) for v in g["gnorm:" + k])
        n64, m64 = (float(v) for v in g64["gnorm:" + k])
        scale = _grad_scale(g64, k, m64, top)
        pos = sample_positions(gr.numel(), 16, zlib_crc(k))
        assert np.array_equal(pos, g["gpos:" + k]) and np.array_equal(pos, g64["gpos:" + k])
        e_hip = float(np.abs(gr[pos].numpy() - g64["gval:" + k]).max()) / scale
        e_ref = float(np.abs(g["gval:" + k] - g64["gval:" + k]).max()) / scale
        rows.append((k, e_hip, e_ref))
        for name, v in (("hip_entry", e_hip), ("ref_entry", e_ref)):
            if v > worst[name][0]:
                worst[name] = (v, k)
        assert e_hip < max(3.0 * e_ref, 1e-4), (k, e_hip, e_ref)
        if m64 > 5e-4 * top:
            n_hip = abs(float(gr.double().norm()) - n64) / n64
            n_ref = abs(ref_norm - n64) / n64
            for name, v in (("hip_norm", n_hip), ("ref_norm", n_ref)):
                if v > worst[name][0]:
                    worst[name] = (v, k)
            assert n_hip < max(3.0 * n_ref, 1e-4), (k, n_hip, n_ref)
    rows.sort(key=lambda r: -r[2])
    print(f"cfg2 128^3 [{precision}] against the fp64 reference: worst gradient-norm error HIP "
          f"{worst['hip_norm'][0]:.2e} ({worst['hip_norm'][1]}) / reference fp32 "
          f"{worst['ref_norm'][0]:.2e} ({worst['ref_norm'][1]}); worst sampled entry HIP "
          f"{worst['hip_entry'][0]:.2e} ({worst['hip_entry'][1]}) / reference fp32 "
          f"{worst['ref_entry'][0]:.2e} ({worst['ref_entry'][1]})")
    for k, e_hip, e_ref in rows[:5]:
        print(f"    {k}: HIP {e_hip:.2e}, reference fp32 {e_ref:.2e} of the gradient's max")


def test_cfg2_256x256x128_matches_reference(cuda):
    g = np.load(os.path.join(GOLD, "unet3d_cfg2_full_256x256x128.npz"))
    shape = tuple(int(v) for v in g["shape"])
    x, y = full_inputs(shape, int(g["seed"]))
    assert abs(float(x.double().sum()) - g["x_checksum"][0]) < 1e-6 * g["x_checksum"][0]
    net = build(cuda)
    prob = net(x.to(cuda))[0]
    with torch.no_grad():
        logits = net(x.to(cuda), return_logits=True)[0]
    _check_forward(g, logits, "cfg2 256x256x128 [f16x3]")
    del logits
    loss = compound_loss(prob, y.to(cuda))
    np.testing.assert_allclose(float(loss.detach()), float(g["loss"]), rtol=1e-4)
    if "grad_keys" not in g.files:
        return
    loss.backward()
    top = max(float(g["gnorm:" + str(k)][1]) for k in g["grad_keys"])
    for k, p in net.named_parameters():
        if ("gnorm:" + k) not in g.files:
            continue
        gr = p.grad.detach().float().cpu().reshape(-1)
        ref_norm, ref_max = (float(v) for v in g["gnorm:" + k])
        scale = _grad_scale(g, k, ref_max, top)
        assert float(np.abs(gr[g["gpos:" + k]].numpy() - g["gval:" + k]).max()) / scale < 2e-3, k
        if ref_max > 5e-4 * top:
            assert abs(float(gr.double().norm()) - ref_norm) / ref_norm < 1e-3, k


def test_f16x3_equals_fp32_mfma_at_two_128_cubed(cuda):
    """The default (f16x3 split) path against the bit-exact fp32-MFMA path at the benchmark's
    batch: logits, and batch item 0 against the one-item run (items are independent)."""
    x = torch.cat([full_inputs((1, 2, 128, 128, 128), s)[0] for s in (1, 2)]).to(cuda)
    net = build(cuda)
    old = HF.CONV_PRECISION
    try:
        out = {}
        for mode in ("f16x3", "fp32"):
            HF.set_conv_precision(mode)
            with torch.no_grad():
                out[mode] = net(x, return_logits=True)[0]
        HF.set_conv_precision("f16x3")
        with torch.no_grad():
            one = net(x[:1], return_logits=True)[0]
    finally:
        HF.set_conv_precision(old)
    rng = float(out["fp32"].max() - out["fp32"].min())
    err = float((out["f16x3"] - out["fp32"]).abs().max()) / rng
    print(f"f16x3 vs fp32-MFMA logits at 2x128^3: {err:.2e} of the logit range")
    assert err < 2e-5
    # one item alone vs the same item inside a batch of two: other launch decompositions
    # (split-K, statistics fold), same arithmetic
    assert float((one - out["f16x3"][:1]).abs().max()) / rng < 2e-6


SHAPE = (24, 20, 28)   # 3 x 2.5 x 3.5 bricks of 8: interior, face, edge and ragged bricks


@pytest.mark.parametrize("cin,cout", [(32, 32), (64, 64), (64, 32), (32, 64)])
def test_spec_instances_and_zring_against_c_oracle(cuda, cin, cout):
    """Forward (SPEC = 3 for 32 output channels, SPEC = 1 for 64), backward-data (the same
    kernels on flipped weights) and the z-ring backward-weight kernel vs fp64-accumulating C
    loops (oracle/c/adell_oracle.c) on NCDHW arrays."""
    rng = np.random.default_rng(cin * 100 + cout)
    x = rng.standard_normal((1, cin, *SHAPE), dtype=np.float32)
    w = (rng.standard_normal((cout, cin, 3, 3, 3), dtype=np.float32) * 0.05).astype(np.float32)
    b = rng.standard_normal((cout,), dtype=np.float32)
    dy = rng.standard_normal((1, cout, *SHAPE), dtype=np.float32)
    y_ref = cops.conv3d(x, w, b, 1, 1)
    dx_ref, dw_ref, db_ref = cops.conv3d_bwd(x, w, dy, 1, 1)
    xt, wt = ops.ndhwc(torch.from_numpy(x).to(cuda)), torch.from_numpy(w).to(cuda)
    dyt, bt = ops.ndhwc(torch.from_numpy(dy).to(cuda)), torch.from_numpy(b).to(cuda)
    for nospec in (0, 1):
        with _lib.tuning(igemm_nospec=nospec):
            y, _ = ops.conv3d_fwd(xt, ops.pack_weight_f16x3(wt, 0), bt, cout, 3, 1, 1)
            dx, _ = ops.conv3d_bwd_data(dyt, ops.pack_weight_f16x3(wt, 1), SHAPE, cin, 0, 3, 1, 1)
        e_y = np.abs(y.cpu().numpy() - y_ref).max() / np.abs(y_ref).max()
        e_dx = np.abs(dx.cpu().numpy() - dx_ref).max() / np.abs(dx_ref).max()
        assert e_y < 2e-6 and e_dx < 2e-6, (nospec, e_y, e_dx)
    dw, db = ops.conv3d_bwd_weight(xt, dyt, 3, 1, 1, want_db=True, f16x3=True)
    e_dw = np.abs(dw.view(cout, cin, 3, 3, 3).cpu().numpy() - dw_ref).max() / np.abs(dw_ref).max()
    e_db = np.abs(db.cpu().numpy() - db_ref).max() / np.abs(db_ref).max()
    assert e_dw < 5e-6 and e_db < 5e-6, (e_dw, e_db)
