import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adell_mri_amd import ops, functional as HF
dev = torch.device("cuda:0")
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for rows, K, N in [(65536, 96, 384), (8192, 192, 768), (1024, 384, 1536), (128, 768, 3072), (16, 1024, 2048), (432, 768, 3072)]:
    x = torch.randn(rows, K, device=dev); w = torch.randn(N, K, device=dev); dy = torch.randn(rows, N, device=dev)
    t_f = timeit(lambda: ops.gemm(rows, N, K, x, K, True, w, K, True))
    t_d = timeit(lambda: ops.gemm(rows, K, N, dy, N, True, w, K, False))
    t_w = timeit(lambda: ops.gemm(N, K, rows, dy, N, False, x, K, False))
    t_b = timeit(lambda: ops.bias_grad(HF._rows_as_volume(dy)))
    fl = 2.0 * rows * K * N
    print(f"rows={rows} {K}->{N}: fwd {t_f:7.1f} us ({fl/t_f/1e6:6.1f} TF) dgrad {t_d:7.1f} us ({fl/t_d/1e6:6.1f} TF) wgrad {t_w:7.1f} us ({fl/t_w/1e6:6.1f} TF) db {t_b:7.1f} us", flush=True)
