"""Backward-data of the 2 -> 32 input conv (dx has 2 channels): MFMA igemm (N tile 94 % empty)
against the vector-ALU small-Cin kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adell_mri_amd import ops
dev = torch.device("cuda:0")
sz, batch, cin, cout = 128, 2, 2, 32
w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
dy = ops.ndhwc(torch.randn(batch, cout, sz, sz, sz, device=dev))
wpb = ops.pack_weight_f16x3(w, 1)
def t(fn):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 30 * 1e3
a = t(lambda: ops.conv3d_bwd_data(dy, wpb, (sz,) * 3, cin, 0, 3, 1, 1))
b = t(lambda: ops.conv_cin_small_bwd_data(dy, w, (sz,) * 3, (1, 1, 1)))
r0 = ops.conv3d_bwd_data(dy, wpb, (sz,) * 3, cin, 0, 3, 1, 1)[0]
r1 = ops.conv_cin_small_bwd_data(dy, w, (sz,) * 3, (1, 1, 1))
print(f"igemm {a:.1f} us, cin_small {b:.1f} us, rel diff {float((r0 - r1).abs().max() / r1.abs().max()):.2e}")
