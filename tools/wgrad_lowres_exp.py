"""Weight-gradient time of low-resolution, wide 3x3x3 layers with the z-ring kernel and with the
per-plane kernel (adell_set_tuning wgrad_nozring): python tools/wgrad_lowres_exp.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from adell_mri_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
L = _lib.lib()


def timed(fn, reps=20):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for C, size in ((32, (128, 128, 128)), (64, (128, 128, 128)), (64, (64, 64, 64)), (128, (32, 32, 32)),
                (512, (9, 9, 33)), (512, (17, 17, 65)), (256, (17, 17, 65)), (128, (33, 33, 65)),
                (128, (65, 65, 65)), (256, (8, 8, 8)), (128, (16, 16, 16))):
    N = 2 if size[0] in (8, 16, 32, 64, 128) else 1
    x = ops.ndhwc(torch.randn(N, C, *size, device=dev))
    dy = ops.ndhwc(torch.randn(N, C, *size, device=dev))
    flops = 2.0 * N * size[0] * size[1] * size[2] * C * C * 27
    out = []
    for nz in (0, 1):
        assert L.adell_set_tuning(b"wgrad_nozring", nz) == 0
        us = timed(lambda: ops.conv3d_bwd_weight(x, dy, 3, 1, 1, want_db=True, f16x3=True))
        out.append(f"{'per-plane' if nz else 'z-ring'} {us:8.1f} us {flops / us / 1e6:6.1f} TF")
    L.adell_set_tuning(b"wgrad_nozring", 0)
    print(f"{C}->{C} at {size} batch {N}: " + " | ".join(out))
