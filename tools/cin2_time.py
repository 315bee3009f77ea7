"""Time of the Conv3d(2 -> 2) forward of the U-Net input block at 2 x 128^3 (67 MB in + out)."""
import sys
import torch
sys.path.insert(0, ".")
from adell_mri_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
x = ops.ndhwc(torch.randn(2, 2, 128, 128, 128, device=dev, generator=g))
w = torch.randn(2, 2, 3, 3, 3, device=dev, generator=g)
b = torch.randn(2, device=dev, generator=g)
fn = lambda: ops.conv_cin_small_fwd(x, w, b, (1, 1, 1), True)
for _ in range(5): fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): fn()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 50 * 1e3
print("cin_small_fwd 2->2 @2x128^3:", round(us, 1), "us", round(2 * x.numel() * 4 / us / 1e6, 2), "TB/s")
