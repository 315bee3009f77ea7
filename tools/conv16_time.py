"""16 -> 16 channel 3^3 conv at UNETR's full resolution (4 x 96^3): the z-marching 16-column kernel
(csrc/conv_zring16.hip) against the implicit-GEMM instance it replaces; fp32 and split-row sources,
forward (bias + statistics) and backward-data. Alternating windows in one process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adell_mri_amd import _lib, ops
dev = torch.device("cuda:0")
N, sz = (int(sys.argv[1]) if len(sys.argv) > 1 else 4), (int(sys.argv[2]) if len(sys.argv) > 2 else 96)
g = torch.Generator().manual_seed(1)
x = ops.ndhwc(torch.randn(N, 16, sz, sz, sz, generator=g).to(dev))
w = (torch.randn(16, 16, 3, 3, 3, generator=g) * 0.05).to(dev)
b = torch.randn(16, generator=g).to(dev)
wf, wb = ops.pack_weight_f16x3(w, 0), ops.pack_weight_f16x3(w, 1)
rows, sr = ops.rows_from_f32(x, 9)
amax = torch.zeros(1, device=dev, dtype=torch.int32)


def timed(f, n=10):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


cases = {
    "fwd fp32 (+bias, stats, absmax)": lambda: ops.conv3d_fwd(x, wf, b, 16, 3, 1, 1, want_stats=True, amax=amax),
    "fwd rows (+bias, stats)": lambda: ops.conv3d_fwd(rows, wf, b, 16, 3, 1, 1, want_stats=True, rows0=sr),
    "dgrad": lambda: ops.conv3d_bwd_data(x, wb, (sz, sz, sz), 16, 0, 3, 1, 1, amax=amax),
    "dgrad + add": lambda: ops.conv3d_bwd_data(x, wb, (sz, sz, sz), 16, 0, 3, 1, 1, amax=amax, add0=x),
}
flop = 2.0 * N * sz ** 3 * 16 * 16 * 27
byts = N * sz ** 3 * 32 * 4
for name, f in cases.items():
    new, old = [], []
    for _ in range(3):
        new.append(timed(f))
        with _lib.tuning(igemm_no16=1):
            old.append(timed(f))
    a, o = min(new), min(old)
    print(f"{name:34s} z-ring16 {a:.3f} ms ({flop / a / 1e9:.0f} TF, {byts / a / 1e6:.0f} GB/s algorithmic)   "
          f"implicit GEMM {o:.3f} ms ({flop / o / 1e9:.0f} TF)", flush=True)
