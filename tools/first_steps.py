"""Wall time of each of the first N training steps of a fresh process (memory-pool growth, stream
creation, first-use packing): python tools/first_steps.py [N=40]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from adell_mri_amd.parallel import GradSync  # noqa: E402
from adell_mri_amd.trainer import StepRunner  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0")
net, _ = bench.build_module(dev, bench.CONFIG)
net.train()
opt = net.configure_optimizers()["optimizer"]
runner = StepRunner(net, opt, GradSync(opt))
batch = bench.synthetic_batch(2, (128, 128, 128), dev, 42)
ms = []
for i in range(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    runner.train_step(batch)
    torch.cuda.synchronize()
    ms.append(round((time.perf_counter() - t0) * 1e3, 1))
print(json.dumps({"ms": ms, "reserved_GB": round(torch.cuda.memory_reserved() / 2**30, 2),
                  "allocated_peak_GB": round(torch.cuda.max_memory_allocated() / 2**30, 2)}))
