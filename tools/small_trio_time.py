"""Times the HBM-bound layers of the bench step one kernel at a time, through ops.py, on whatever
build ADELL_HIP_LIBRARY points to (compare builds by running it once per build on one box):
the 32 -> 32 stride-2 trio (fused forward / backward-data with and without the fork operand /
weight gradient), the 2 -> 32 narrow-input trio (forward / dW / dX) and the weight repack.
Prints one JSON line; `frac` = algorithmic bytes / time / 8 TB/s.
    python tools/small_trio_time.py [edge=128] [batch=2]"""
import json
import os
import sys

import torch

sys.path.insert(0, ".")
from adell_mri_amd import functional as HF  # noqa: E402
from adell_mri_amd import ops  # noqa: E402

edge = int(sys.argv[1]) if len(sys.argv) > 1 else 128
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda:0")
size = (edge,) * 3
half = (edge // 2,) * 3


def timed(fn, reps=40, warm=60):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


out = {"lib": os.path.basename(os.environ.get("ADELL_HIP_LIBRARY", "libadellhip.so")),
       "edge": edge, "batch": batch}


def put(name, us, nbytes):
    out[name] = {"us": round(us, 1), "frac": round(nbytes / us / 1e6 / 8.0, 3)}


g = torch.Generator(device=dev).manual_seed(1)
big = 4 * batch * 32 * edge ** 3          # bytes of a 32-channel full-resolution tensor
small = big // 8

# ---- stride-2 trio ----------------------------------------------------------------------------
w = torch.randn(32, 32, 3, 3, 3, device=dev, generator=g) * 0.05
x = ops.ndhwc(torch.randn(batch, 32, *size, device=dev, generator=g))
dy = ops.ndhwc(torch.randn(batch, 32, *half, device=dev, generator=g))
add0 = ops.ndhwc(torch.randn(batch, 32, *size, device=dev, generator=g))
wf, wb = HF._packed(w, 0), HF._packed(w, 1)
amax = torch.zeros(2, dtype=torch.int32, device=dev)
put("s2_fwd", timed(lambda: ops.conv3d_fwd(x, wf, None, 32, (3, 3, 3), (2, 2, 2), (1, 1, 1),
                                           want_stats=True, amax=amax[0:1])), big + small)
put("s2_dgrad", timed(lambda: ops.conv3d_bwd_data_s2_fused(dy, wb, size, amax=amax[1:2])),
    big + small)
put("s2_dgrad_add0", timed(lambda: ops.conv3d_bwd_data_s2_fused(dy, wb, size, amax=amax[1:2],
                                                                add0=add0)), 2 * big + small)
put("s2_wgrad", timed(lambda: ops.conv3d_bwd_weight(x, dy, (3, 3, 3), (2, 2, 2), (1, 1, 1),
                                                    want_db=True, f16x3=True, x_amax=amax[0:1],
                                                    dy_amax=amax[1:2])), big + small)
del x, add0

# ---- narrow-input trio (2 -> 32) ---------------------------------------------------------------
w2 = torch.randn(32, 2, 3, 3, 3, device=dev, generator=g) * 0.1
x2 = ops.ndhwc(torch.randn(batch, 2, *size, device=dev, generator=g))
dy2 = ops.ndhwc(torch.randn(batch, 32, *size, device=dev, generator=g))
b2 = torch.randn(32, device=dev, generator=g)
thin = big // 16
put("cin2_fwd", timed(lambda: ops.conv_cinfold_fwd(x2, w2, b2, (1, 1, 1), True)), big + thin)
try:
    put("cin2_fwd_f16x3", timed(lambda: ops.conv_cinfold_fwd(x2, w2, b2, (1, 1, 1), True, f16x3=True)),
        big + thin)
except (TypeError, AttributeError):
    pass     # an older build / binding without the split-f16 forward
put("cin2_wgrad", timed(lambda: ops.conv_cinfold_bwd_weight(x2, dy2, (1, 1, 1), True)), big + thin)
put("cin2_dx", timed(lambda: ops.conv_cinfold_bwd_data(dy2, w2, size, (1, 1, 1))), big + thin)
try:
    put("cin2_dx_f16x3", timed(lambda: ops.conv_cinfold_bwd_data(dy2, w2, size, (1, 1, 1), f16x3=True)),
        big + thin)
except (TypeError, AttributeError):
    pass
print(json.dumps(out))
