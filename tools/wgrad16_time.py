"""Timing of the 16 x 16 tile form of the z-ring weight gradient (conv_wgrad_zring.hip) against the
32 x 32 tile form on UNETR's full-resolution layers (4 x 96^3). Alternating windows in one process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adell_mri_amd import _lib, ops
dev = torch.device("cuda:0")
N, sz = 4, 96
for c0, c1, cout in [(16, 0, 16), (32, 0, 16), (16, 16, 16), (16, 0, 32)]:
    g = torch.Generator().manual_seed(1)
    x0 = ops.ndhwc(torch.randn(N, c0, sz, sz, sz, generator=g).to(dev))
    x1 = ops.ndhwc(torch.randn(N, c1, sz, sz, sz, generator=g).to(dev)) if c1 else None
    dy = ops.ndhwc((torch.randn(N, cout, sz, sz, sz, generator=g) * 1e-3).to(dev))
    xa = torch.maximum(x0.abs().max(), x1.abs().max() if c1 else x0.abs().max()).view(1).view(torch.int32)
    ya = dy.abs().max().view(1).view(torch.int32)
    f = lambda: ops.conv3d_bwd_weight(x0, dy, 3, 1, 1, x1=x1, want_db=True, f16x3=True, x_amax=xa, dy_amax=ya)

    def timed(n=10):
        for _ in range(2):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            f()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    res = {"t16": [], "t32": []}
    for _ in range(3):
        res["t16"].append(timed())
        with _lib.tuning(wgrad_no16=1):
            res["t32"].append(timed())
    flop = 2 * N * sz ** 3 * (c0 + c1) * cout * 27
    byts = N * sz ** 3 * (c0 + c1 + cout) * 4
    with _lib.tuning(zr_oldseg=1):
        old = min(timed() for _ in range(2))
    a, b = min(res["t16"]), min(res["t32"])
    print(f"   (16x16 tiles with the round-2 segment rule: {old:.3f} ms)")
    print(f"wgrad {c0}+{c1}->{cout} @ {N}x{sz}^3: 16x16 tiles {a:.3f} ms ({flop / a / 1e9:.0f} TF, "
          f"{byts / a / 1e6:.0f} GB/s algorithmic)  32x32 tiles {b:.3f} ms ({flop / b / 1e9:.0f} TF)", flush=True)

# the 32 x 32 form on shapes whose column count does not fill the blocks evenly
for n, c, sz in [(4, 32, 96), (2, 32, 128), (4, 64, 48), (1, 32, 128), (1, 32, 96)]:
    g = torch.Generator().manual_seed(1)
    x0 = ops.ndhwc(torch.randn(n, c, sz, sz, sz, generator=g).to(dev))
    dy = ops.ndhwc((torch.randn(n, c, sz, sz, sz, generator=g) * 1e-3).to(dev))
    xa = x0.abs().max().view(1).view(torch.int32)
    ya = dy.abs().max().view(1).view(torch.int32)
    f = lambda: ops.conv3d_bwd_weight(x0, dy, 3, 1, 1, want_db=True, f16x3=True, x_amax=xa, dy_amax=ya)

    def timed2(k=10):
        for _ in range(2):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(k):
            f()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / k

    new = min(timed2() for _ in range(3))
    with _lib.tuning(zr_oldseg=1):
        old = min(timed2() for _ in range(3))
    flop = 2 * n * sz ** 3 * c * c * 27
    print(f"wgrad {c}->{c} @ {n}x{sz}^3: balanced segments {new:.3f} ms ({flop / new / 1e9:.0f} TF)  "
          f"round-2 rule {old:.3f} ms ({flop / old / 1e9:.0f} TF)", flush=True)
