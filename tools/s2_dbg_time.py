"""Where the time of the one-launch stride-2 kernels goes: each kernel with parts switched off
(debug build only: ADELL_HIP_LIBRARY=adell_mri_amd/libadellhip_dbg.so, made with
`make ADELL_DEBUG=1`; results are wrong when a switch is on). Bits: 1 no MFMAs, 2 no stores /
statistics, 8 no split + LDS stores, 16 no halo loads after the first phase.
    python tools/s2_dbg_time.py [edge=128] [batch=2] [fwd,dgrad,wgrad]"""
import json
import sys

import torch

sys.path.insert(0, ".")
from adell_mri_amd import _lib  # noqa: E402
from adell_mri_amd import functional as HF  # noqa: E402
from adell_mri_amd import ops  # noqa: E402

edge = int(sys.argv[1]) if len(sys.argv) > 1 else 128
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda:0")
size, half = (edge,) * 3, (edge // 2,) * 3


def timed(fn, reps=200, warm=100):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / reps * 1e3, 1)


g = torch.Generator(device=dev).manual_seed(1)
w = torch.randn(32, 32, 3, 3, 3, device=dev, generator=g) * 0.05
x = ops.ndhwc(torch.randn(batch, 32, *size, device=dev, generator=g))
dy = ops.ndhwc(torch.randn(batch, 32, *half, device=dev, generator=g))
wf, wb = HF._packed(w, 0), HF._packed(w, 1)
amax = torch.zeros(2, dtype=torch.int32, device=dev)
kernels = {
    "fwd": ("igemm_dbg", (0, 1, 2, 8, 16, 3, 27, 0), lambda: ops.conv3d_fwd(x, wf, None, 32, (3, 3, 3), (2, 2, 2), (1, 1, 1),
                                                want_stats=True, amax=amax[0:1])),
    "dgrad": ("igemm_dbg", (0, 1, 2, 4, 8, 16, 3, 27, 0), lambda: ops.conv3d_bwd_data_s2_fused(dy, wb, size, amax=amax[1:2])),
    "wgrad": ("zr_dbg", (0, 1, 8, 16, 25, 0), lambda: ops.conv3d_bwd_weight(x, dy, (3, 3, 3), (2, 2, 2), (1, 1, 1),
                                                      want_db=True, f16x3=True, x_amax=amax[0:1],
                                                      dy_amax=amax[1:2])),
}
out = {}
only = sys.argv[3].split(",") if len(sys.argv) > 3 else list(kernels)
for name, (switch, modes, fn) in kernels.items():
    if name not in only:
        continue
    row = {}
    for dbg in modes:
        _lib.lib().adell_set_tuning(switch.encode(), dbg)
        row.setdefault(str(dbg), []).append(timed(fn))
    _lib.lib().adell_set_tuning(switch.encode(), 0)
    out[name] = row
print(json.dumps(out))
