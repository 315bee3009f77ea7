"""MFMA-bound regime of the f16x3 GEMM (big square problems): TF/s sustained over back-to-back
launches (warm clocks). usage: gemm_big_exp.py [n=4096] [seconds=2]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from adell_mri_amd import ops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
dev = torch.device("cuda:0")
A = torch.randn(n, n, device=dev)
B = torch.randn(n, n, device=dev)
out = torch.empty(n, n, device=dev)
for name, fn in (("f16x3", lambda: ops.gemm_f16x3(n, n, n, A, n, True, B, n, True, out=out)),
                 ("fp32", lambda: ops.gemm(n, n, n, A, n, True, B, n, True, out=out))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.time()
    reps = 0
    best = None
    while time.time() - t0 < secs:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        best = ms if best is None else min(best, ms)
        last = ms
        reps += 10
    print(f"{name}: n={n} last window {last:.3f} ms = {2 * n ** 3 / last / 1e9:.0f} TF, best {2 * n ** 3 / best / 1e9:.0f} TF ({reps} launches)")
