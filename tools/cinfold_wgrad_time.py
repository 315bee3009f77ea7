"""Weight gradient of the 2 -> 32 input conv at 2 x 128^3: split-f16 MFMA kernel against the fp32-MFMA
kernel (csrc/conv_cinfold.hip), alternating windows."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adell_mri_amd import ops
dev = torch.device("cuda:0")
for N, cin, cout, sz in [(2, 2, 32, 128), (4, 1, 16, 96), (1, 2, 32, 128)]:
    g = torch.Generator().manual_seed(1)
    x = ops.ndhwc(torch.randn(N, cin, sz, sz, sz, generator=g).to(dev))
    dy = ops.ndhwc((torch.randn(N, cout, sz, sz, sz, generator=g) * 1e-3).to(dev))

    def timed(f16, n=10):
        f = lambda: ops.conv_cinfold_bwd_weight(x, dy, (1, 1, 1), True, f16x3=f16)
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

    a, b = [], []
    for _ in range(3):
        a.append(timed(True)); b.append(timed(False))
    byts = (x.numel() + dy.numel()) * 4
    print(f"wgrad {cin}->{cout} @ {N}x{sz}^3: f16x3 {min(a):.3f} ms ({byts / min(a) / 1e6:.0f} GB/s)  "
          f"fp32 MFMA {min(b):.3f} ms ({byts / min(b) / 1e6:.0f} GB/s)", flush=True)
