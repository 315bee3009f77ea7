"""Phase timeline of the f16x3 implicit-GEMM blocks from in-kernel s_memtime stamps (debug build:
bash tools/build_dbg.sh; ADELL_HIP_LIBRARY=adell_mri_amd/libadellhip_dbg.so). For a forward conv
Cin -> Cout at edge^3 (one item): per block the cycles from kernel-side start to each phase
boundary -- chunk c: start / halo loads arrived + scale known / halo image written / taps issued --
then epilogue start, stores issued, end. Medians over the blocks, and the launch's wall time.
    python tools/igemm_stamps.py [cin=32] [cout=32] [edge=128]"""
import ctypes
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from adell_mri_amd import _lib, ops  # noqa: E402

cin = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cout = int(sys.argv[2]) if len(sys.argv) > 2 else 32
edge = int(sys.argv[3]) if len(sys.argv) > 3 else 128
dev = torch.device("cuda:0")
x = ops.ndhwc(torch.randn(1, cin, edge, edge, edge, device=dev))
w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
b = torch.randn(cout, device=dev)
wp = ops.pack_weight_f16x3(w, 0)
for _ in range(20):
    ops.conv3d_fwd(x, wp, b, cout, 3, 1, 1, want_stats=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.conv3d_fwd(x, wp, b, cout, 3, 1, 1, want_stats=True)
e1.record()
torch.cuda.synchronize()
wall_us = e0.elapsed_time(e1) * 100.0
nblk = 1 << 16
buf = torch.zeros(nblk * 24, dtype=torch.int64, device=dev)
lib = _lib.lib()
lib.adell_debug_set_stamps.argtypes = [ctypes.c_void_p]
assert lib.adell_debug_set_stamps(ctypes.c_void_p(buf.data_ptr())) == 0
ops.conv3d_fwd(x, wp, b, cout, 3, 1, 1, want_stats=True)
torch.cuda.synchronize()
lib.adell_debug_set_stamps(ctypes.c_void_p(0))
s = buf.view(nblk, 24).cpu().numpy()
used = s[:, 0] != 0
s = s[used]
names = {0: "start"}
nchunk = (cin + 15) // 16
for c in range(min(nchunk, 4)):
    names[1 + 4 * c] = f"chunk{c} begin"
    names[2 + 4 * c] = f"chunk{c} halo arrived, scale known"
    names[3 + 4 * c] = f"chunk{c} halo image written"
names[20], names[21], names[22] = "taps of the last chunk issued", "stores issued", "end"
out = {"layer": f"fwd {cin}->{cout} @ {edge}^3", "blocks": int(used.sum()), "launch_us": round(wall_us, 1)}
prev = 0
rows = []
for k in sorted(names):
    rel = s[:, k] - s[:, 0]
    med = float(statistics.median(rel.tolist()))
    rows.append((names[k], int(med), int(med - prev)))
    prev = med
out["median_cycles_from_start (phase delta)"] = rows
# blocks resident together on a CU start at about the same stamp: spread of the start stamps
out["block_life_cycles_median"] = int(statistics.median((s[:, 22] - s[:, 0]).tolist()))
rows_ = out.pop("median_cycles_from_start (phase delta)")
print(json.dumps(out))
for name, at, delta in rows_:
    print(f"{at:9d} (+{delta:7d})  {name}")
