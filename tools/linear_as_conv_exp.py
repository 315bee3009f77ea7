"""Linear layers (ConvNeXt pointwise MLPs, ViT / SWIN projections) through the f16x3 1x1x1 conv kernels
against the fp32-MFMA GEMM: forward + backward time and the error against fp64, per shape.
    python tools/linear_as_conv_exp.py"""
import json
import sys

import torch

sys.path.insert(0, ".")
from adell_mri_amd import functional as HF  # noqa: E402
from adell_mri_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
# (rows, in, out): VICReg ConvNeXt-3D stages at 64^3 crops, 32 items; UNETR tokens (4 x 216 x 768)
SHAPES = [(131072, 96, 384), (131072, 384, 96), (16384, 192, 768), (16384, 768, 192),
          (2048, 384, 1536), (2048, 1536, 384), (256, 768, 3072), (256, 3072, 768),
          (864, 768, 768), (864, 768, 3072), (864, 3072, 768)]


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
    return e0.elapsed_time(e1) / reps


for rows, cin, cout in SHAPES:
    if rows % 64:
        rows_ = rows // 64 * 64
    else:
        rows_ = rows
    g = torch.Generator().manual_seed(rows + cin)
    x = torch.randn(rows_, cin, generator=g)
    w = torch.randn(cout, cin, generator=g) / cin ** 0.5
    b = torch.randn(cout, generator=g)
    dy = torch.randn(rows_, cout, generator=g)
    xd, wd, bd, dyd = (t.to(dev) for t in (x, w, b, dy))

    def gemm_step():
        xg, wg, bg = xd.clone().requires_grad_(True), wd.clone().requires_grad_(True), bd.clone().requires_grad_(True)
        y = HF.linear(xg, wg, bg)
        y.backward(dyd)
        return y, xg.grad, wg.grad, bg.grad

    def conv_step():
        xg, wg, bg = xd.clone().requires_grad_(True), wd.clone().requires_grad_(True), bd.clone().requires_grad_(True)
        y = HF.conv3d(as_volume(xg), wg.view(cout, cin, 1, 1, 1), bg, 1, 0, want_stats=False)
        y.backward(as_volume(dyd))
        return y.permute(0, 2, 3, 4, 1).reshape(rows_, cout), xg.grad, wg.grad, bg.grad

    ref_y = (x.double() @ w.double().t() + b.double())
    ref_dx = dy.double() @ w.double()
    ref_dw = dy.double().t() @ x.double()
    out = {"rows": rows_, "in": cin, "out": cout}
    for name, fn in (("gemm", gemm_step), ("conv", conv_step)):
        y, dx, dw, db = fn()
        rel = lambda a, r: float((a.detach().cpu().double() - r).abs().max() / r.abs().max())
        out[name + "_ms"] = round(timed(fn), 4)
        out[name + "_err"] = [float("%.2g" % rel(y, ref_y)), float("%.2g" % rel(dx, ref_dx)),
                              float("%.2g" % rel(dw, ref_dw))]
    flops = 2.0 * rows_ * cin * cout * 3
    out["gemm_TF"] = round(flops / out["gemm_ms"] / 1e9, 1)
    out["conv_TF"] = round(flops / out["conv_ms"] / 1e9, 1)
    print(json.dumps(out), flush=True)
