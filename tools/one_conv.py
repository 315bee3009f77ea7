"""Run ONE conv kernel shape a few times (for rocprofv3 --pmc runs).
usage: one_conv.py <fwd|dgrad|wgrad> Cin Cout size [k] [stride]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adell_mri_amd import ops
kind, cin, cout, sz = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
k = int(sys.argv[5]) if len(sys.argv) > 5 else 3
s = int(sys.argv[6]) if len(sys.argv) > 6 else 1
dev = torch.device("cuda:0")
x = ops.ndhwc(torch.randn(1, cin, sz, sz, sz, device=dev))
w = torch.randn(cout, cin, k, k, k, device=dev) * 0.05
b = torch.randn(cout, device=dev)
wp, wpb = ops.pack_weight_f16x3(w, 0), ops.pack_weight_f16x3(w, 1)
y, _ = ops.conv3d_fwd(x, wp, b, cout, k, s, k // 2, want_stats=True)
dy = torch.randn_like(y)
for _ in range(3):
    if kind == "fwd":
        ops.conv3d_fwd(x, wp, b, cout, k, s, k // 2, want_stats=True)
    elif kind == "dgrad":
        ops.conv3d_bwd_data(dy, wpb, (sz,) * 3, cin, 0, k, s, k // 2)
    else:
        ops.conv3d_bwd_weight(x, dy, k, s, k // 2, f16x3=True)
torch.cuda.synchronize()
