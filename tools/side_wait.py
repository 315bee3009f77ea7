"""How long the main stream waits for the weight-gradient stream at the end of the backward pass
(functional.FLAGS['wgrad_stream']): events around the join, per step.
    python tools/side_wait.py [steps=12]"""
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from adell_mri_amd import functional as HF  # noqa: E402
from adell_mri_amd.parallel import GradSync  # noqa: E402
from adell_mri_amd.trainer import StepRunner  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
HF.FLAGS["wgrad_stream"] = True
dev = torch.device("cuda:0")
net, _ = bench.build_module(dev, bench.CONFIG)
net.train()
opt = net.configure_optimizers()["optimizer"]
runner = StepRunner(net, opt, GradSync(opt))
batch = bench.synthetic_batch(int(net.batch_size), (128, 128, 128), dev, 42)
pairs = []
real = HF.join_side_stream


def timed_join():
    if not HF._SIDE["pending"]:
        return real()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    real()
    e1.record()
    pairs.append((e0, e1))


HF.join_side_stream = timed_join
HF._side_join_callback.__globals__["join_side_stream"] = timed_join
for _ in range(6):
    runner.train_step(batch)
torch.cuda.synchronize()
pairs.clear()
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(steps):
    runner.train_step(batch)
t1.record()
torch.cuda.synchronize()
waits = [a.elapsed_time(b) for a, b in pairs]
print(json.dumps({"ms_per_step": round(t0.elapsed_time(t1) / steps, 3), "joins": len(waits),
                  "wait_ms_median": round(statistics.median(waits), 3),
                  "wait_ms_max": round(max(waits), 3)}))
