"""Timing of the f16x3 backward-weight kernels at the big layers of BASELINE config 2."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adell_mri_amd import ops
dev = torch.device("cuda:0")
for cin, cout, sz in [(32, 32, 128), (64, 64, 128), (32, 32, 64), (64, 64, 64), (128, 128, 32), (64, 64, 32),
                      (64, 64, 16), (128, 128, 16), (256, 256, 16), (128, 128, 8)]:
    x = ops.ndhwc(torch.randn(1, cin, sz, sz, sz, device=dev))
    dy = ops.ndhwc(torch.randn(1, cout, sz, sz, sz, device=dev) * 1e-3)
    xa = x.abs().max().view(1).view(torch.int32)
    ya = dy.abs().max().view(1).view(torch.int32)
    f = lambda: ops.conv3d_bwd_weight(x, dy, 3, 1, 1, want_db=True, f16x3=True, x_amax=xa, dy_amax=ya)
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"wgrad {cin}->{cout}@{sz}: {ms:.3f} ms {2 * sz ** 3 * cin * cout * 27 / ms / 1e9:.0f} TF "
          f"(incl. slab reduce)", flush=True)
