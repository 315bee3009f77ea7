#!/usr/bin/env python
"""Per-layer view of the headline step (bench.py workload): every conv launch grouped by
(kernel, direction, shape) with its time share and algorithmic TFLOP/s."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from adell_mri_amd import ops  # noqa: E402
from adell_mri_amd.parallel import GradSync  # noqa: E402
from adell_mri_amd.trainer import StepRunner  # noqa: E402

dev = torch.device("cuda", 0)
size = int(sys.argv[1]) if len(sys.argv) > 1 else 128
net, _ = bench.build_module(dev)
net.train()
opt = net.configure_optimizers()["optimizer"]
runner = StepRunner(net, opt, GradSync(opt))
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 2
batch = bench.synthetic_batch(nb, size, dev, 42)
for _ in range(2):
    runner.train_step(batch)
torch.cuda.synchronize()
ops.KERNEL_TIMER = ops.KernelTimer()
steps = 3
for _ in range(steps):
    runner.train_step(batch)
tags = ops.KERNEL_TIMER.by_tag()
ops.KERNEL_TIMER = None
tot = sum(v["ms"] for v in tags.values())
print(f"timed kernels: {tot / steps:.2f} ms/step")
for (name, tag), v in sorted(tags.items(), key=lambda kv: -kv[1]["ms"]):
    print(f"{v['ms'] / steps:7.3f} ms {100 * v['ms'] / tot:5.1f}% {v['tflops']:7.1f} TF x{v['launches'] // steps:2d}  "
          f"{name.replace('adell_', '')}  {tag}")
