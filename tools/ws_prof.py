"""Role cycle census of the persistent wave-specialised conv kernel (needs a -DADELL_DEBUG build:
make -C adell_mri_amd/csrc clean all ADELL_DEBUG=1 OUT=../libadellhip_dbg.so is NOT used; this
script builds its own debug library into gpurun_out/). usage: ws_prof.py Cin Cout size"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from adell_mri_amd import _lib, ops
cin, cout, sz = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda:0")
x = ops.ndhwc(torch.randn(1, cin, sz, sz, sz, device=dev))
w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
b = torch.randn(cout, device=dev)
wp = ops.pack_weight_f16x3(w, 0)
_lib.lib().adell_set_tuning(b"igemm_ws", 1)
for _ in range(3):
    ops.conv3d_fwd(x, wp, b, cout, 3, 1, 1, want_stats=True)
h = _lib.lib()
h.adell_set_tuning(b"igemm_ws", 1)
out = (ctypes.c_ulonglong * 16)()
h.adell_debug_ws_prof(out)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
ops.conv3d_fwd(x, wp, b, cout, 3, 1, 1, want_stats=True)
e1.record()
torch.cuda.synchronize()
h.adell_debug_ws_prof(out)
v = list(out)
nb = max(v[5], 1)
print(f"{cin}->{cout}@{sz}: {e0.elapsed_time(e1):.3f} ms, blocks {v[5]}")
print(f"compute wave: stage bodies {v[0]/nb:.0f}  barrier waits {v[1]/nb:.0f}  epilogues {v[2]/nb:.0f} cycles per block")
print("loader work by stage index g:", [round(v[8 + q] / nb) for q in range(4)])
print(f"loader wave:  work {v[3]/nb:.0f}  barrier waits {v[4]/nb:.0f} cycles per block")
