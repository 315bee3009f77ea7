#!/usr/bin/env python
"""Host enqueue time vs GPU time of the headline step: if the host needs nearly as long to
enqueue a step as the GPU needs to run it, launch overhead is (close to) the critical path."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from adell_mri_amd.parallel import GradSync  # noqa: E402
from adell_mri_amd.trainer import StepRunner  # noqa: E402

dev = torch.device("cuda", 0)
net = bench.build_module(dev, 128)
net.train()
opt = net.configure_optimizers()["optimizer"]
runner = StepRunner(net, opt, GradSync(opt))
batch = bench.synthetic_batch(1, 128, dev, 42)
for _ in range(3):
    runner.train_step(batch)
torch.cuda.synchronize()
n = 10
cpu = 0.0
t0 = time.perf_counter()
for _ in range(n):
    a = time.perf_counter()
    runner.train_step(batch)
    cpu += time.perf_counter() - a
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"host enqueue {1e3 * cpu / n:.2f} ms/step, wall {1e3 * t_all / n:.2f} ms/step "
      f"(enqueue loop finished after {1e3 * t_enq / n:.2f} ms/step)")
# host-only cost: same loop with the GPU idle at the start of every step
cpu2 = 0.0
for _ in range(5):
    torch.cuda.synchronize()
    a = time.perf_counter()
    runner.train_step(batch)
    cpu2 += time.perf_counter() - a
print(f"host enqueue with an idle GPU {1e3 * cpu2 / 5:.2f} ms/step")
