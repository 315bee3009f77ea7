"""Parity-class backward-data of a stride-2 3^3 conv under each forced tile configuration."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adell_mri_amd import _lib, ops
from adell_mri_amd import functional as HF
cin, cout, sz, batch = (int(v) for v in sys.argv[1:5])
dev = torch.device("cuda:0")
w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
dy = ops.ndhwc(torch.randn(batch, cout, sz // 2, sz // 2, sz // 2, device=dev))
packs = HF._packed_s2_classes(w, (1, 1, 1))
def run():
    return ops.conv3d_bwd_data_s2(dy, packs, (sz,) * 3, cin, (1, 1, 1))
for _ in range(100):
    run()
torch.cuda.synchronize()
for cfg in (-1, 0, 1, 2, 3, 6, -1):
    _lib.lib().adell_debug_force_conv_cfg(cfg)
    try:
        for _ in range(5): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        print(f"cfg {cfg:2d}: {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us")
    except Exception as exc:  # noqa: BLE001
        print(f"cfg {cfg:2d}: {str(exc)[:90]}")
_lib.lib().adell_debug_force_conv_cfg(-1)
