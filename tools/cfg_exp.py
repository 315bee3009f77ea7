"""Forward time of one conv shape under each forced tile configuration (adell_debug_force_conv_cfg).
usage: cfg_exp.py Cin Cout size|DxHxW batch [k] [stride]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adell_mri_amd import _lib, ops
cin, cout, batch = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[4])
dims = [int(v) for v in sys.argv[3].split("x")]
dims = dims * 3 if len(dims) == 1 else dims
k = int(sys.argv[5]) if len(sys.argv) > 5 else 3
s = int(sys.argv[6]) if len(sys.argv) > 6 else 1
dev = torch.device("cuda:0")
x = ops.ndhwc(torch.randn(batch, cin, *dims, device=dev))
w = torch.randn(cout, cin, k, k, k, device=dev) * 0.05
b = torch.randn(cout, device=dev)
wp = ops.pack_weight_f16x3(w, 0)
flops = 2.0 * batch * (dims[0] // s) * (dims[1] // s) * (dims[2] // s) * cin * cout * k ** 3
for _ in range(300):   # clock ramp: the first ~100 ms of load run slow
    ops.conv3d_fwd(x, wp, b, cout, k, s, k // 2, want_stats=True)
torch.cuda.synchronize()
for cfg in (-1, 0, 1, 2, 3, 6, -1):
    _lib.lib().adell_debug_force_conv_cfg(cfg)
    try:
        for _ in range(5):
            ops.conv3d_fwd(x, wp, b, cout, k, s, k // 2, want_stats=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            ops.conv3d_fwd(x, wp, b, cout, k, s, k // 2, want_stats=True)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 30 * 1e3
        print(f"cfg {cfg:2d}: {us:8.1f} us  {flops / us / 1e6:6.1f} TF")
    except Exception as exc:  # noqa: BLE001
        print(f"cfg {cfg:2d}: {str(exc)[:100]}")
_lib.lib().adell_debug_force_conv_cfg(-1)
