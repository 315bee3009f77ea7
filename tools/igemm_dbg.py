"""Timing decomposition of the f16x3 implicit-GEMM kernel (ADELL_IGEMM_DBG experiments).
usage: igemm_dbg.py  (reads ADELL_IGEMM_DBG from the environment; results are wrong when set)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adell_mri_amd import ops
dev = torch.device("cuda:0")
for cin, cout, sz in [(32, 32, 128), (64, 64, 128), (64, 32, 128)]:
    x = ops.ndhwc(torch.randn(1, cin, sz, sz, sz, device=dev))
    w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
    b = torch.randn(cout, device=dev)
    wp = ops.pack_weight_f16x3(w, 0)
    for _ in range(2):
        ops.conv3d_fwd(x, wp, b, cout, 3, 1, 1, want_stats=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.conv3d_fwd(x, wp, b, cout, 3, 1, 1, want_stats=True)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"dbg={os.environ.get('ADELL_IGEMM_DBG', '0')} fwd {cin}->{cout}@{sz}: {ms:.3f} ms "
          f"{2 * sz ** 3 * cin * cout * 27 / ms / 1e9:.0f} TF", flush=True)
