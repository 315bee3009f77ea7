"""Weight gradient of the big Linear layers of config 4 (dW = dY^T X over 262 144 / 32 768 / 4 096 rows):
the f16x3 GEMM (outer / outer operands, split-K slabs) against the conv weight-gradient kernels run on
the same tensors viewed as a 1x1x1 convolution over a [rows / 64, 8, 8] volume. us per call."""
import json
import sys

import torch

sys.path.insert(0, ".")
from adell_mri_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
ops.FLAGS["gemm_f16x3"] = True


def as_volume(t2):      # [rows, C] -> [1, C, rows / 64, 8, 8] view with NDHWC memory
    rows, C = t2.shape
    return t2.view(1, rows // 64, 8, 8, C).permute(0, 4, 1, 2, 3)


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


out = {}
g = torch.Generator(device=dev).manual_seed(0)
for rows, cin, cout in ((262144, 96, 384), (262144, 384, 96), (32768, 192, 768), (32768, 768, 192),
                        (4096, 384, 1536), (4096, 1536, 384)):
    x = torch.randn(rows, cin, device=dev, generator=g)
    dy = torch.randn(rows, cout, device=dev, generator=g)
    nbytes = 4 * rows * (cin + cout)
    t_gemm = timed(lambda: ops.gemm_f16x3(cout, cin, rows, dy, cout, False, x, cin, False))
    xv, dv = as_volume(x), as_volume(dy)
    t_conv = timed(lambda: ops.conv3d_bwd_weight(xv, dv, (1, 1, 1), (1, 1, 1), (0, 0, 0), want_db=False,
                                                 f16x3=True))
    ref = dy.double().t() @ x.double()
    a = ops.gemm_f16x3(cout, cin, rows, dy, cout, False, x, cin, False)
    b = ops.conv3d_bwd_weight(xv, dv, (1, 1, 1), (1, 1, 1), (0, 0, 0), want_db=False, f16x3=True)
    b = b[0] if isinstance(b, (tuple, list)) else b
    ea = float((a.double() - ref).abs().max() / ref.abs().max())
    eb = float((b.reshape(cout, cin).double() - ref).abs().max() / ref.abs().max())
    out[f"{rows}x{cin}->{cout}"] = {"gemm_us": round(t_gemm, 1), "gemm_TBs": round(nbytes / t_gemm / 1e6, 2),
                                    "conv_us": round(t_conv, 1), "conv_TBs": round(nbytes / t_conv / 1e6, 2),
                                    "err_gemm": ea, "err_conv": eb}
print(json.dumps(out))
