"""Per-layer HIP-event times of the config-2b training step as bench.py builds it
(python tools/cfg2b_layers.py [steps]): every timed launch by (kernel family, shape tag)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from adell_mri_amd import ops  # noqa: E402
from adell_mri_amd.trainer import StepRunner  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
net, _ = bench.build_cfg2b(dev)
net.train()
batch = bench.synthetic_batch(1, (128, 128, 128), dev, 1)
runner = StepRunner(net)
for _ in range(2):
    runner.train_step(batch)
torch.cuda.synchronize()
ops.KERNEL_TIMER = ops.KernelTimer()
for _ in range(steps):
    runner.train_step(batch)
torch.cuda.synchronize()
timer, ops.KERNEL_TIMER = ops.KERNEL_TIMER, None
tags = timer.by_tag()
tot = sum(v["ms"] for v in tags.values())
print(f"timed kernels {tot / steps:.2f} ms/step")
for (name, tag), v in sorted(tags.items(), key=lambda kv: -kv[1]["ms"])[:70]:
    print(f"{v['ms'] / steps:7.3f} ms {100 * v['ms'] / tot:5.1f}% {v['tflops']:7.1f} TF "
          f"x{v['launches'] // steps:2d}  {name.replace('adell_', '')}  {tag}")
