"""Backward-data of the 32 -> 32 stride-2 layer: one persistent launch (csrc/conv_dgrad_s2.hip) against
the eight parity-class launches, with and without the fork operand. Warm clocks, 40 launches each.
    python tools/dgrad_s2_fused_exp.py [edge=128] [batch=2]"""
import json
import sys

import torch

sys.path.insert(0, ".")
from adell_mri_amd import functional as HF  # noqa: E402
from adell_mri_amd import ops  # noqa: E402

edge = int(sys.argv[1]) if len(sys.argv) > 1 else 128
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda:0")
size = (edge,) * 3
w = (torch.randn(32, 32, 3, 3, 3, device=dev) * 0.05)
dy = ops.ndhwc(torch.randn(batch, 32, edge // 2, edge // 2, edge // 2, device=dev))
add0 = ops.ndhwc(torch.randn(batch, 32, *size, device=dev))
full, classes = HF._packed(w, 1), HF._packed_s2_classes(w, (1, 1, 1))


def timed(fn, reps=40):
    for _ in range(60):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


out = {"edge": edge, "batch": batch}
for name, add in (("plain", None), ("add0", add0)):
    out[f"fused_{name}_us"] = round(timed(lambda: ops.conv3d_bwd_data_s2_fused(dy, full, size, add0=add)), 1)
    out[f"classes_{name}_us"] = round(
        timed(lambda: ops.conv3d_bwd_data_s2(dy, classes, size, 32, (1, 1, 1), add0=add)), 1)
print(json.dumps(out))
