"""The three GEMMs of a Linear layer, each on the fp32-MFMA GEMM and on the f16x3 conv kernels
(1x1x1 conv forward / backward-data / backward-weight with cached packs and absmax words)."""
import json
import sys

import torch

sys.path.insert(0, ".")
from adell_mri_amd import functional as HF  # noqa: E402
from adell_mri_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
SHAPES = [(131072, 96, 384), (131072, 384, 96), (16384, 192, 768), (16384, 768, 192),
          (2048, 384, 1536), (2048, 1536, 384), (256, 768, 3072), (832, 768, 3072), (832, 3072, 768)]


def vol(t2):
    rows, C = t2.shape
    return t2.view(1, rows // 64, 8, 8, C).permute(0, 4, 1, 2, 3)


def timed(fn, reps=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for rows, cin, cout in SHAPES:
    x = torch.randn(rows, cin, device=dev)
    w = torch.randn(cout, cin, device=dev) / cin ** 0.5
    dy = torch.randn(rows, cout, device=dev)
    w5 = w.view(cout, cin, 1, 1, 1)
    p0, p1 = HF._packed(w5, 0), HF._packed(w5, 1)
    xa = x.abs().max().view(1).view(torch.int32)
    ya = dy.abs().max().view(1).view(torch.int32)
    wa = w.abs().max().view(1).view(torch.int32)
    size = (rows // 64, 8, 8)
    fl = 2.0 * rows * cin * cout
    out = {"rows": rows, "in": cin, "out": cout}
    t = {
        "gemm_fwd": timed(lambda: ops.gemm(rows, cout, cin, x, cin, True, w, cin, True)),
        "gemm_dx": timed(lambda: ops.gemm(rows, cin, cout, dy, cout, True, w, cin, False)),
        "gemm_dw": timed(lambda: ops.gemm(cout, cin, rows, dy, cout, False, x, cin, False)),
        "h_fwd": timed(lambda: ops.gemm_f16x3(rows, cout, cin, x, cin, True, w, cin, True)),
        "h_dx": timed(lambda: ops.gemm_f16x3(rows, cin, cout, dy, cout, True, w, cin, False)),
        "h_dw": timed(lambda: ops.gemm_f16x3(cout, cin, rows, dy, cout, False, x, cin, False)),
        "amax_x": timed(lambda: ops.absmax_word(x)),
        "h_fwd_words": timed(lambda: ops.gemm_f16x3(rows, cout, cin, x, cin, True, w, cin, True, xa, wa)),
        "conv_fwd": timed(lambda: ops.conv3d_fwd(vol(x), p0, None, cout, 1, 1, 0, want_stats=False)),
        "conv_dx": timed(lambda: ops.conv3d_bwd_data(vol(dy), p1, size, cin, 0, (1, 1, 1), (1, 1, 1), (0, 0, 0))),
        "conv_dw": timed(lambda: ops.conv3d_bwd_weight(vol(x), vol(dy), 1, 1, 0, f16x3=True, x_amax=xa, dy_amax=ya)),
    }
    for k, v in t.items():
        out[k] = f"{v:.0f}us {fl / v / 1e6:.0f}TF"
    print(json.dumps(out), flush=True)
