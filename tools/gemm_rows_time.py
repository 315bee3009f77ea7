"""Times the Linear-layer GEMM shapes of BASELINE config 4 (VICReg ConvNeXt, 64 crops of 64^3: the
point-wise MLPs of res_blocks.py:588-604 at 262 144 / 32 768 / 4 096 / 512 rows) on the streaming
kernel (csrc/gemm_rows.hip) and on the tile kernel (gemm_norows=1), forward (with the GELU pair),
backward-data and plain: us and algorithmic TB/s. One JSON line.
    python tools/gemm_rows_time.py"""
import json
import sys

import torch

sys.path.insert(0, ".")
from adell_mri_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
ops.FLAGS["gemm_f16x3"] = True


def timed(fn, reps=20, warm=5):
    for _ in range(warm):
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
for rows, c in ((262144, 96), (32768, 192), (4096, 384)):
    h = 4 * c
    x = torch.randn(rows, c, device=dev, generator=g)
    hid = torch.randn(rows, h, device=dev, generator=g)
    w1 = torch.randn(h, c, device=dev, generator=g) * 0.05
    w2 = torch.randn(c, h, device=dev, generator=g) * 0.05
    b1, b2 = torch.randn(h, device=dev, generator=g), torch.randn(c, device=dev, generator=g)
    cases = {
        # forward pwconv1 + GELU (reads x, writes pre and post), pwconv2 (reads post, writes y)
        "fwd1_gelu": (lambda: ops.gemm_f16x3_act(rows, h, c, x, c, True, w1, c, True, "gelu", bias=b1,
                                                 want_act=True), 4 * rows * (c + 2 * h)),
        "fwd2": (lambda: ops.gemm_f16x3(rows, c, h, hid, h, True, w2, h, True, bias=b2), 4 * rows * (c + h)),
        # backward-data: dpost = dy W2 (x gelu'(pre)), dx = dpre W1
        "dx2_dgelu": (lambda: ops.gemm_f16x3_act(rows, h, c, x, c, True, w2, h, False, "gelu",
                                                 dact_in=hid), 4 * rows * (c + 2 * h)),
        "dx1": (lambda: ops.gemm_f16x3(rows, c, h, hid, h, True, w1, c, False), 4 * rows * (c + h)),
    }
    for name, (fn, nbytes) in cases.items():
        rec = {}
        for label, sw in (("rows", 0), ("tile", 1)):
            with _lib.tuning(gemm_norows=sw):
                us = timed(fn)
            rec[label] = {"us": round(us, 1), "TBs": round(nbytes / us / 1e6, 2)}
        out[f"{rows}x{c}:{name}"] = rec
print(json.dumps(out))
