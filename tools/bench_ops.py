"""Micro-benchmark of the conv kernels at BASELINE config-2 layer shapes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adell_mri_amd import ops

dev = torch.device("cuda:0")
SHAPES = [  # Cin(C0,C1), size, Cout, k, s, p
    ((64, 0), 128, 64, 3, 1, 1),
    ((32, 32), 128, 64, 3, 1, 1),
    ((64, 0), 128, 32, 3, 1, 1),
    ((32, 0), 128, 32, 3, 1, 1),
    ((2, 0), 128, 32, 3, 1, 1),
    ((32, 0), 128, 32, 3, 2, 1),
    ((64, 0), 32, 64, 3, 1, 1),
    ((128, 0), 16, 128, 3, 1, 1),
    ((256, 0), 8, 256, 3, 1, 1),
]
which = sys.argv[1] if len(sys.argv) > 1 else "fwd,bwd_data,bwd_weight"

def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

for (c0, c1), sz, cout, k, s, p in SHAPES:
    cin = c0 + c1
    x0 = ops.ndhwc(torch.randn(1, c0, sz, sz, sz, device=dev))
    x1 = ops.ndhwc(torch.randn(1, c1, sz, sz, sz, device=dev)) if c1 else None
    w = torch.randn(cout, cin, k, k, k, device=dev) * 0.05
    b = torch.randn(cout, device=dev)
    if "f16" in which:
        wp, wpb = ops.pack_weight_f16x3(w, 0), ops.pack_weight_f16x3(w, 1)
    else:
        wp, wpb = ops.pack_weight(w, 0), ops.pack_weight(w, 1)
    y, _ = ops.conv3d_fwd(x0, wp, b, cout, k, s, p, x1=x1, want_stats=True)
    osz = y.shape[2]
    flops = 2.0 * osz ** 3 * cout * cin * k ** 3
    line = f"Cin={cin:3d} Cout={cout:3d} {sz}^3 s{s}: {flops/1e9:7.1f} GF"
    if "fwd" in which:
        t = timeit(lambda: ops.conv3d_fwd(x0, wp, b, cout, k, s, p, x1=x1, want_stats=True))
        line += f" | fwd {t:7.3f} ms {flops/t/1e9:6.1f} TF"
    dy = torch.randn_like(y)
    if "bwd_data" in which:
        t = timeit(lambda: ops.conv3d_bwd_data(dy, wpb, (sz,) * 3, c0, c1, k, s, p))
        line += f" | dX {t:7.3f} ms {flops/t/1e9:6.1f} TF"
    if "bwd_weight" in which:
        t = timeit(lambda: ops.conv3d_bwd_weight(x0, dy, k, s, p, x1=x1, f16x3=("f16" in which)))
        line += f" | dW {t:7.3f} ms {flops/t/1e9:6.1f} TF"
    print(line, flush=True)
