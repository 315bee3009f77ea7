"""Does a conv kernel hold its rate under sustained load (DVFS)? Times windows of 50 back-to-back
launches for ~3 s. usage: sustained.py Cin Cout size [ws]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adell_mri_amd import _lib, ops
cin, cout, sz = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
ws = int(sys.argv[4]) if len(sys.argv) > 4 else 0
dev = torch.device("cuda:0")
x = ops.ndhwc(torch.randn(2, cin, sz, sz, sz, device=dev))
w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
b = torch.randn(cout, device=dev)
wp = ops.pack_weight_f16x3(w, 0)
flops = 2.0 * 2 * sz ** 3 * cin * cout * 27
_lib.lib().adell_set_tuning(b"igemm_ws", ws)
out = []
for win in range(24):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.conv3d_fwd(x, wp, b, cout, 3, 1, 1, want_stats=True)
    e1.record()
    torch.cuda.synchronize()
    out.append(flops / (e0.elapsed_time(e1) / 50) / 1e9)
print(f"{cin}->{cout}@{sz} ws={ws}: TF per 50-launch window:", " ".join(f"{v:.0f}" for v in out))
