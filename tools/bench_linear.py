"""Micro-benchmark of HF.linear (the MFMA conv kernel on [rows, C] operands) at the
Linear shapes of the ConvNeXt / ViT paths: forward and backward time per call."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from adell_mri_amd import functional as HF  # noqa: E402

dev = torch.device("cuda:0")
SHAPES = [(65536, 96, 384), (65536, 384, 96), (8192, 192, 768), (8192, 768, 192),
          (1024, 384, 1536), (1024, 1536, 384), (128, 768, 3072), (128, 3072, 768),
          (16, 768, 1024), (16, 1024, 2048), (16, 2048, 1024),
          (2 * 216, 768, 768), (2 * 216, 768, 3072)]


def timeit(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for rows, cin, cout in SHAPES:
    x = torch.randn(rows, cin, device=dev, requires_grad=True)
    w = (torch.randn(cout, cin, device=dev) * 0.05).requires_grad_(True)
    b = torch.randn(cout, device=dev, requires_grad=True)
    flops = 2.0 * rows * cin * cout
    tf = timeit(lambda: HF.linear(x, w, b))
    y = HF.linear(x, w, b)
    dy = torch.randn_like(y)

    def fb():
        x.grad = w.grad = b.grad = None
        HF.linear(x, w, b).backward(dy)

    tfb = timeit(fb)
    byt = 4.0 * (rows * cin + rows * cout + cin * cout)
    print(f"rows={rows:6d} {cin:5d}->{cout:5d}: {flops/1e9:7.2f} GF | fwd {tf*1e3:8.1f} us "
          f"{flops/tf/1e9:7.1f} TF {byt/tf/1e6:7.1f} GB/s | fwd+bwd {tfb*1e3:8.1f} us "
          f"(bwd {flops*2/(tfb-tf)/1e9:7.1f} TF)", flush=True)
