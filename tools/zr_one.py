import os, sys
sys.path.insert(0, "/root/repo")
import torch
from adell_mri_amd import ops
dev = torch.device("cuda:0")
cin = cout = 64; sz = 128
x = ops.ndhwc(torch.randn(1, cin, sz, sz, sz, device=dev))
dy = ops.ndhwc(torch.randn(1, cout, sz, sz, sz, device=dev) * 1e-3)
xa = x.abs().max().view(1).view(torch.int32); ya = dy.abs().max().view(1).view(torch.int32)
for _ in range(3):
    ops.conv3d_bwd_weight(x, dy, 3, 1, 1, want_db=True, f16x3=True, x_amax=xa, dy_amax=ya)
torch.cuda.synchronize()
