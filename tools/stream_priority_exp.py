"""EXPERIMENT: the training step on a HIGH-priority HIP stream with the weight-gradient side stream
at normal priority (so that the backward-data chain gets the CUs first and the weight gradients
fill in under the HBM-bound passes), against both at normal priority. Alternating blocks of steps
in one process."""
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from adell_mri_amd import functional as HF  # noqa: E402
from adell_mri_amd.parallel import GradSync  # noqa: E402
from adell_mri_amd.trainer import StepRunner  # noqa: E402

dev = torch.device("cuda:0")
print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "n/a")
net, _ = bench.build_module(dev, bench.CONFIG)
net.train()
opt = net.configure_optimizers()["optimizer"]
runner = StepRunner(net, opt, GradSync(opt))
batch = bench.synthetic_batch(2, (128, 128, 128), dev, 42)
hi = torch.cuda.Stream(priority=-1)


def block(n, stream=None):
    ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        runner.train_step(batch)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            runner.train_step(batch)
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for _ in range(4):
    runner.train_step(batch)
with torch.cuda.stream(hi):
    for _ in range(4):
        runner.train_step(batch)
torch.cuda.synchronize()
res = {"normal": [], "main_high": []}
for r in range(4):
    res["normal"].append(round(block(10), 3))
    res["main_high"].append(round(block(10, hi), 3))
print(json.dumps({"ms_per_step": {k: statistics.median(v) for k, v in res.items()}, "blocks": res}))
