import sys, torch
sys.path.insert(0, ".")
from adell_mri_amd import ops
dev = torch.device("cuda:0")
def t(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for cin, cout, sz in ((32, 32, 128), (32, 32, 64), (64, 64, 32)):
    x = ops.ndhwc(torch.randn(2, cin, sz, sz, sz, device=dev))
    w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
    b = torch.randn(cout, device=dev)
    wp = ops.pack_weight_f16x3(w, 0)
    us = t(lambda: ops.conv3d_fwd(x, wp, b, cout, 3, 2, 1, want_stats=True))
    print(f"fwd s2 {cin}->{cout} @ {sz}: {us:.1f} us")
