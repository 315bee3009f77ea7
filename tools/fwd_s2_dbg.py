"""Where the one-launch stride-2 forward kernel spends its time: the -DADELL_DEBUG build of
csrc/conv_fwd_s2.hip (adell_mri_amd/libadellhip_dbg.so, `bash tools/build_dbg.sh`) with phases
switched off (results are wrong then). bits: 1 no MFMAs, 2 no y stores, 8 no halo split, 16 no halo loads after the first phases."""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, ".")
from adell_mri_amd import functional as HF  # noqa: E402
from adell_mri_amd import ops  # noqa: E402

edge = int(sys.argv[1]) if len(sys.argv) > 1 else 128
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 2
here = os.path.dirname(os.path.abspath(__file__))
L = ctypes.CDLL(os.path.join(here, "..", "adell_mri_amd", "libadellhip_dbg.so"))
L.adell_conv3d_fwd_s2_fused.argtypes = [ctypes.c_void_p] * 7 + [ctypes.c_int] + [ctypes.c_void_p] * 2
dev = torch.device("cuda:0")
size = (edge,) * 3
w = torch.randn(32, 32, 3, 3, 3, device=dev) * 0.05
x = ops.ndhwc(torch.randn(batch, 32, *size, device=dev))
y = ops.new_act(batch, 32, edge // 2, edge // 2, edge // 2, dev)
pack = HF._packed(w, 0)
d = ops.make_conv_desc(batch, size, 32, 0, 32, 3, 2, 1)
stream = torch.cuda.current_stream().cuda_stream


def launch():
    rc = L.adell_conv3d_fwd_s2_fused(ctypes.addressof(d), x.data_ptr(), pack.halfs.data_ptr(),
                                     pack.scale.data_ptr(), None, y.data_ptr(), None, 0, None, stream)
    assert rc == 0, rc


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
    return round(e0.elapsed_time(e1) / reps * 1e3, 1)


out = {}
for bits in (0, 1, 2, 3, 8, 16, 27):
    assert L.adell_set_tuning(b"igemm_dbg", bits) == 0
    out[str(bits)] = timed(launch)
print(json.dumps(out))
