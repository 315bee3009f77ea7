"""Timing experiments on the wave-specialised conv kernel (needs the -DADELL_DEBUG library:
results are wrong while igemm_dbg != 0). usage: ws_exp.py Cin Cout size [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adell_mri_amd import _lib, ops
cin, cout, sz = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
nb = int(sys.argv[4]) if len(sys.argv) > 4 else 2
dev = torch.device("cuda:0")
x = ops.ndhwc(torch.randn(nb, cin, sz, sz, sz, device=dev))
w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
b = torch.randn(cout, device=dev)
wp = ops.pack_weight_f16x3(w, 0)
flops = 2.0 * nb * sz ** 3 * cin * cout * 27
def t(label, **tune):
    tune.setdefault("igemm_ws", 1)
    with _lib.tuning(**tune):
        for _ in range(2):
            ops.conv3d_fwd(x, wp, b, cout, 3, 1, 1, want_stats=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.conv3d_fwd(x, wp, b, cout, 3, 1, 1, want_stats=True)
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"{label:44s} {ms:7.3f} ms  {flops / ms / 1e9:7.1f} TF")
print(f"{cin}->{cout} @ {sz}^3 x {nb}")
t("block kernel", igemm_ws=0)
t("ws")
t("ws, loaders: no halo", igemm_dbg=1)
t("ws, loaders: no weights", igemm_dbg=2)
t("ws, loaders: nothing", igemm_dbg=3)
t("ws, compute: no MFMA", igemm_dbg=8)
t("ws, compute: no epilogue", igemm_dbg=16)
t("ws, no MFMA, no epilogue", igemm_dbg=24)
t("ws, loaders nothing, no epilogue", igemm_dbg=19)
t("ws, barriers only", igemm_dbg=27)
t("ws, loaders never wait for DMA", igemm_dbg=32)
t("ws, no halo, no DMA waits", igemm_dbg=33)
