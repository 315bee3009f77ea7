#!/usr/bin/env python
"""Times the launch-shape outliers of the headline step (stride-2 convs, transposed convs,
2-channel layers) one at a time, optionally under a forced tile config.
usage: small_layers.py [cfg ...]   (cfg -1 = the planner's choice)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from adell_mri_amd import _lib, ops  # noqa: E402
from adell_mri_amd import functional as HF  # noqa: E402

dev = torch.device("cuda", 0)
cfgs = [int(a) for a in sys.argv[1:]] or [-1]
B = int(os.environ.get("BATCH", "2"))


def layer(kind, cin, cout, size, k=3, s=1):
    x = ops.ndhwc(torch.randn(B, cin, size, size, size, device=dev)).requires_grad_(True)
    if kind == "convT":
        w = (torch.randn(cin, cout, 2, 2, 2, device=dev) * 0.05).requires_grad_(True)
        b = torch.zeros(cout, device=dev, requires_grad=True)
        return lambda: HF.conv_transpose3d(x, w, b), (x, w, b)
    w = (torch.randn(cout, cin, k, k, k, device=dev) * 0.05).requires_grad_(True)
    b = torch.zeros(cout, device=dev, requires_grad=True)
    return lambda: HF.conv3d(x, w, b, s, k // 2), (x, w, b)


LAYERS = [("conv", 32, 32, 128, 3, 2), ("conv", 32, 32, 64, 3, 2), ("conv", 64, 64, 32, 3, 2),
          ("convT", 32, 32, 64), ("convT", 64, 32, 32), ("convT", 128, 64, 16),
          ("conv", 2, 32, 128, 3, 1), ("conv", 2, 2, 128, 3, 1)]

for cfg in cfgs:
    _lib.lib().adell_debug_force_conv_cfg(cfg)
    print(f"== cfg {cfg}")
    for spec in LAYERS:
        try:
            fn, leaves = layer(*spec)
            y = fn()
            dy = torch.randn_like(y)
            for _ in range(2):
                y = fn()
                y.backward(dy)
            torch.cuda.synchronize()
            ops.KERNEL_TIMER = ops.KernelTimer()
            reps = 5
            for _ in range(reps):
                y = fn()
                y.backward(dy)
            tags = ops.KERNEL_TIMER.by_tag()
            ops.KERNEL_TIMER = None
            for (name, tag), v in sorted(tags.items(), key=lambda kv: -kv[1]["ms"]):
                print(f"  {v['ms'] / reps:7.3f} ms {v['tflops']:7.1f} TF  {name.replace('adell_', '')}  {tag}")
            del y, dy, leaves, fn
        except Exception as exc:  # a forced config may not fit the layer
            ops.KERNEL_TIMER = None
            print(f"  {spec}: {type(exc).__name__}: {str(exc)[:100]}")
    _lib.lib().adell_debug_force_conv_cfg(-1)
