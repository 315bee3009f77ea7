"""Per-launch HIP-event times of a secondary config's step by (kernel family, shape tag), one stream:
ADELL_BENCH_ONLY=<key> python tools/secondary_layers.py [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from adell_mri_amd import functional as HF  # noqa: E402
from adell_mri_amd import ops  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2


def probe(key, runner, batch):
    overlap = HF.FLAGS["wgrad_stream"]
    HF.FLAGS["wgrad_stream"] = False
    runner.train_step(batch)
    torch.cuda.synchronize()
    ops.KERNEL_TIMER = ops.KernelTimer()
    for _ in range(steps):
        runner.train_step(batch)
    torch.cuda.synchronize()
    timer, ops.KERNEL_TIMER = ops.KERNEL_TIMER, None
    HF.FLAGS["wgrad_stream"] = overlap
    tags = timer.by_tag()
    tot = sum(v["ms"] for v in tags.values())
    print(f"==== {key}: timed kernels {tot / steps:.2f} ms/step")
    for (name, tag), v in sorted(tags.items(), key=lambda kv: -kv[1]["ms"])[:60]:
        print(f"{v['ms'] / steps:7.3f} ms {100 * v['ms'] / tot:5.1f}% {v['tflops']:7.1f} TF "
              f"x{v['launches'] // steps:3d}  {name.replace('adell_', '')}  {tag}")
    return {}


bench.PROBE = probe
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
bench.other_config_runs(dev, 0, 1, torch.cuda.synchronize, lambda v, d: float(v))
