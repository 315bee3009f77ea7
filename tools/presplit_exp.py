"""EXPERIMENT: what would storing activations pre-split buy the f16x3 forward kernels?
(VERDICT r1 item 4 i). Converts the input once with adell_presplit (not timed), then alternates
windows of back-to-back launches of the normal kernel and of the same kernel staging pre-split rows
(no block absmax, no fp32 -> hi/lo VALU work), on one box, long enough for the clock to settle.
Needs the experiment build of the library: make -C adell_mri_amd/csrc clean all EXPERIMENTS=1
(rebuild without the flag afterwards). usage: presplit_exp.py [C0 C1 Cout size batch]..."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from adell_mri_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
L = _lib.lib()
vp, i32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_long
L.adell_presplit.argtypes = [vp, i32, i64, i32, vp, vp, vp]
L.adell_presplit.restype = i32
L.adell_conv3d_fwd_f16x3_presplit.argtypes = [ctypes.POINTER(_lib.ConvDesc)] + [vp] * 10
L.adell_conv3d_fwd_f16x3_presplit.restype = i32


def presplit(x):
    """(xs, xk) of an NDHWC-memory activation [N, C, D, H, W]."""
    N, C = x.shape[:2]
    V = x.numel() // (N * C)
    rows = x.permute(0, 2, 3, 4, 1).reshape(N, V, C // 16, 16)
    mx = rows.abs().amax(dim=(1, 3))                              # [N, nchunk]
    e = torch.frexp(mx)[1] - 1                                    # floor(log2(max))
    k = 8 * torch.div(13 - e, 8, rounding_mode="floor")
    k = torch.where(mx > 0, k, torch.zeros_like(k)).clamp(-96, 96).to(torch.int32).contiguous()
    xs = torch.empty(x.numel(), device=x.device, dtype=torch.float32)
    _lib.check(L.adell_presplit(x.data_ptr(), N, V, C, k.data_ptr(), xs.data_ptr(), None))
    return xs, k


def run(c0, c1, cout, sz, batch):
    g = torch.Generator(device="cpu").manual_seed(1)
    x0 = ops.ndhwc((torch.randn(batch, c0, sz, sz, sz, generator=g) * 1.5).to(dev))
    x1 = ops.ndhwc(torch.randn(batch, c1, sz, sz, sz, generator=g).to(dev)) if c1 else None
    w = (torch.randn(cout, c0 + c1, 3, 3, 3, generator=g) * 0.05).to(dev)
    b = torch.randn(cout, generator=g).to(dev)
    wp = ops.pack_weight_f16x3(w, 0)
    y_ref, part_ref = ops.conv3d_fwd(x0, wp, b, cout, 3, 1, 1, x1=x1, want_stats=True)
    xs0, k0 = presplit(x0)
    xs1, k1 = presplit(x1) if c1 else (None, None)
    xk = (k0 if k1 is None else torch.cat([k0, k1], 1)).contiguous()
    d = ops.make_conv_desc(batch, (sz, sz, sz), c0, c1, cout, 3, 1, 1)
    y = ops.new_act(batch, cout, sz, sz, sz, dev)
    part = torch.empty_like(part_ref)
    halfs, scale = wp.halfs, wp.scale

    def launch_pre():
        _lib.check(L.adell_conv3d_fwd_f16x3_presplit(
            ctypes.byref(d), xs0.data_ptr(), None if xs1 is None else xs1.data_ptr(),
            xk.data_ptr(), halfs.data_ptr(), scale.data_ptr(), b.data_ptr(), None, y.data_ptr(),
            part.data_ptr(), None))

    def launch_ref():
        ops.conv3d_fwd(x0, wp, b, cout, 3, 1, 1, x1=x1, want_stats=True)

    launch_pre()
    torch.cuda.synchronize()
    err = float((y - y_ref).abs().max() / y_ref.abs().max())
    perr = float((part - part_ref).abs().max() / part_ref.abs().max())
    flops = 2.0 * batch * sz ** 3 * (c0 + c1) * cout * 27
    res = {"ref": [], "pre": []}
    for win in range(10):
        for name, fn in (("ref", launch_ref), ("pre", launch_pre)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(40):
                fn()
            e1.record()
            torch.cuda.synchronize()
            res[name].append(flops / (e0.elapsed_time(e1) / 40) / 1e9)
    med = {k: sorted(v[3:])[len(v[3:]) // 2] for k, v in res.items()}
    print(f"{c0}+{c1}->{cout} @ {sz}^3 x{batch}: normal {med['ref']:.0f} TF, pre-split input "
          f"{med['pre']:.0f} TF ({med['pre'] / med['ref']:.3f}x); max rel diff y {err:.2e} "
          f"stats {perr:.2e}")


if __name__ == "__main__":
    args = [int(v) for v in sys.argv[1:]]
    cases = [args[i:i + 5] for i in range(0, len(args), 5)] or [
        [32, 0, 32, 128, 2], [64, 0, 64, 128, 2], [32, 32, 32, 128, 2], [64, 0, 64, 64, 2]]
    for c in cases:
        run(*c)
