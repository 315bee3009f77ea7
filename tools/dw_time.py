"""Per-launch time of the depthwise 7^3 forward / backward-data at ConvNeXt's stage-1 shape
(64 crops x 96 channels x 16^3), MFMA Toeplitz form against the vector-ALU kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adell_mri_amd import _lib, ops

dev = torch.device("cuda:0")
shapes = [(64, 96, 16, 16, 16), (64, 384, 4, 4, 4), (16, 96, 32, 16, 16)]
for N, C, D, H, W in shapes:
    x = ops.ndhwc(torch.randn(N, C, D, H, W, device=dev))
    w = torch.randn(C, 1, 7, 7, 7, device=dev) * 0.05
    b = torch.randn(C, device=dev)
    res = {}
    for nomfma in (0, 1):
        with _lib.tuning(dw_nomfma=nomfma):
            for _ in range(3):
                ops.dwconv3d_fwd(x, w, b)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.dwconv3d_fwd(x, w, b)
            e1.record()
            torch.cuda.synchronize()
            res[nomfma] = e0.elapsed_time(e1) / 20 * 1e3
    dy = ops.ndhwc(torch.randn(N, C, D, H, W, device=dev))
    wres = {}
    for nomfma in (0, 1):
        with _lib.tuning(dw_wgrad_nomfma=nomfma):
            for _ in range(3):
                ops.dwconv3d_bwd_weight(x, dy, (7, 7, 7), True)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.dwconv3d_bwd_weight(x, dy, (7, 7, 7), True)
            e1.record()
            torch.cuda.synchronize()
            wres[nomfma] = e0.elapsed_time(e1) / 20 * 1e3
    print(f"{(N, C, D, H, W)}: weight gradient MFMA {wres[0]:7.1f} us   vector ALU {wres[1]:7.1f} us")
    gf = 2.0 * N * C * D * H * W * 343 / 1e9
    print(f"{(N, C, D, H, W)}: MFMA {res[0]:7.1f} us ({gf / res[0] * 1e-3:5.1f} TF)   vector ALU {res[1]:7.1f} us ({gf / res[1] * 1e-3:5.1f} TF)")
