"""Phase switches of the stride-2 weight-gradient kernel (-DADELL_DEBUG build of csrc/conv_wgrad_s2.hip
+ api.hip -> adell_mri_amd/csrc/_dbg/libws2dbg.so; results are wrong when a bit is set): 1 no MFMAs,
8 no split / LDS stores, 16 no loads after the first phase. Times the kernel alone (no slab fold)."""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, ".")
from adell_mri_amd import ops  # noqa: E402

edge = int(sys.argv[1]) if len(sys.argv) > 1 else 128
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 2
here = os.path.dirname(os.path.abspath(__file__))
L = ctypes.CDLL(os.path.join(here, "..", "adell_mri_amd", "csrc", "_dbg", "libws2dbg.so"))


class Plan(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in "ntx nty ntz nbricks blocks R".split()]


dev = torch.device("cuda:0")
x = ops.ndhwc(torch.randn(batch, 32, edge, edge, edge, device=dev))
dy = ops.ndhwc(torch.randn(batch, 32, edge // 2, edge // 2, edge // 2, device=dev))
xa, ya = ops.absmax_word(x), ops.absmax_word(dy)
p = Plan()
i = ctypes.c_int
L.adell_wgrad_s2_plan.argtypes = [i] * 19 + [ctypes.c_void_p]
assert L.adell_wgrad_s2_plan(batch, edge, edge, edge, 32, 0, 32, 3, 3, 3, 2, 2, 2, 1, 1, 1, edge // 2,
                             edge // 2, edge // 2, ctypes.addressof(p)) == 1
slabs = torch.empty(p.R * 27 * 32 * 32 + p.R * 32, device=dev)
vp = ctypes.c_void_p
L.adell_wgrad_s2_launch.argtypes = [vp, i, i, i, i, vp, i, i, i, vp, vp, vp, vp, vp, vp]
st = torch.cuda.current_stream().cuda_stream


def launch():
    rc = L.adell_wgrad_s2_launch(ctypes.addressof(p), batch, edge, edge, edge, x.data_ptr(), edge // 2,
                                 edge // 2, edge // 2, dy.data_ptr(), slabs.data_ptr(), None,
                                 xa.data_ptr(), ya.data_ptr(), st)
    assert rc == 0


def timed(fn, reps=30):
    for _ in range(40):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / reps * 1e3, 1)


out = {"blocks": p.blocks, "bricks": p.nbricks}
for bits in (0, 1, 8, 16, 25):
    assert L.adell_set_tuning(b"zr_dbg", bits) == 0
    out[str(bits)] = timed(launch)
print(json.dumps(out))
