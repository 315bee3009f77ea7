"""ADELL_IGEMM_DBG timing decomposition for the stride-2 and transposed-conv launches.
usage: igemm_dbg_s2.py  (reads ADELL_IGEMM_DBG; results are wrong when set)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adell_mri_amd import ops
dev = torch.device("cuda:0")


def timeit(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10


dbg = os.environ.get("ADELL_IGEMM_DBG", "0")
x = ops.ndhwc(torch.randn(2, 32, 128, 128, 128, device=dev))
w = torch.randn(32, 32, 3, 3, 3, device=dev) * 0.05
b = torch.randn(32, device=dev)
wp = ops.pack_weight_f16x3(w, 0)
print(f"dbg={dbg} fwd s2 32->32@128: {timeit(lambda: ops.conv3d_fwd(x, wp, b, 32, 3, 2, 1, want_stats=True)):.3f} ms", flush=True)
x2 = ops.ndhwc(torch.randn(2, 32, 64, 64, 64, device=dev))
wt = torch.randn(32, 32, 2, 2, 2, device=dev) * 0.05
wpt = ops.pack_weight_f16x3(wt.reshape(32, 32 * 8, 1, 1, 1).contiguous(), 0)
print(f"dbg={dbg} fwd 1x1 32->256@64 (convT-like, plain store): "
      f"{timeit(lambda: ops.conv3d_fwd(x2, wpt, None, 256, 1, 1, 0, want_stats=False)):.3f} ms", flush=True)
